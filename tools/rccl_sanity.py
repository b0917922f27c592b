"""1-GPU sanity check of the RCCL path bench.py / train.py use at N > 1 (the 8-GPU runs belong to the driver): process-group
bring-up with backend nccl (= RCCL), all_reduce / barrier / all_gather on the device, GradBucket round trip."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402
from dkt_amd import distributed  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl")
t = torch.arange(8, device="cuda", dtype=torch.float32)
dist.all_reduce(t)
dist.barrier()
out = [torch.empty_like(t)]
dist.all_gather(out, t)
p = [torch.nn.Parameter(torch.randn(5, device="cuda")), torch.nn.Parameter(torch.randn(3, 4, device="cuda"))]
for q in p:
    q.grad = torch.ones_like(q)
b = distributed.GradBucket(p)
b.allreduce_mean()
torch.cuda.synchronize()
print("RCCL ok: backend", dist.get_backend(), "world", dist.get_world_size(), "all_reduce", t.tolist()[:3], "bucket grads", p[0].grad.tolist()[:2])
dist.destroy_process_group()
