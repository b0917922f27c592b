"""One box, one call: what the headline's Gram forward does next to what the box can do.  Polls rocm-smi (clocks, power, temperatures; ~20 Hz) while (a) the cfg2 Gram forward,
(b) a linear read of the same bytes (torch sum), (c) the Gram backward loop for 4 s each, and prints kernel time + every sampled quantity's median / range.
python tools/box_probe.py"""
import json
import os
import statistics
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
b, n, d = 8192, 105, 1600
z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
w = torch.randn(b, n, n, generator=g, device=dev)
w = (w + w.transpose(1, 2)).contiguous()


def num(v):
    try:
        return float(str(v).strip("()MhzWwCc% "))
    except ValueError:
        return None


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=5)
            j = json.loads(r.stdout)["card0"]
            out.append({k: num(v) for k, v in j.items() if num(v) is not None})
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.03)


for name, fn, nbytes in (("dkt_gram_f32 (cfg2, 8192 episodes)", lambda: ops.gram(z, None, ops.KERNEL_LINEAR_UNIT), b * (n * d + n * n) * 4),
                         ("linear read of Z (torch sum)", lambda: z.sum(), b * n * d * 4),
                         ("dkt_gram_bwd_f32", lambda: ops.gram_bwd(w, z, None, unit_rows=True, w_symmetric=True), b * (2 * n * d + n * n) * 4),
                         ("dkt_gram_f32 again", lambda: ops.gram(z, None, ops.KERNEL_LINEAR_UNIT), b * (n * d + n * n) * 4)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, samples))
    th.start()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 0
    t0.record()
    tw = time.perf_counter()
    while time.perf_counter() - tw < 4.0:
        for _ in range(20):
            fn()
        reps += 20
        torch.cuda.synchronize()
    t1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = t0.elapsed_time(t1) / reps
    print("%s: %.4f ms per call = %.2f TB/s of algorithmic bytes; %d samples" % (name, ms, nbytes / ms / 1e9, len(samples)), flush=True)
    keys = sorted({k for s in samples for k in s})
    for k in keys:
        v = [s[k] for s in samples[2:] if k in s]
        if v and (max(v) > 0):
            print("      %-60s median %8.1f   min %8.1f   max %8.1f" % (k[:60], statistics.median(v), min(v), max(v)), flush=True)
