"""One cfg2 episode per step (B = 1): the kernels of the level-1 training step (Gram -> MLL -> Gram backward + the torch glue), for rocprofv3 --kernel-trace --stats.
Measurement tooling.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b1 -- python tools/b1_step_trace.py [B]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
step, _ = bench._workload("cfg2", b, dev, 0, True)
for _ in range(50):
    step()
torch.cuda.synchronize()
s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(1000):
    step()
t.record()
torch.cuda.synchronize()
print("eager: %.4f ms per step at B = %d" % (s.elapsed_time(t) / 1000, b))
print("hipGraph: %.4f ms per step" % bench._graphed_step_ms("cfg2", b, dev, 0, True))
