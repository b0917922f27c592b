"""Shader clock and socket power the box sustains UNDER each kernel of the headline step (rocm-smi polled from a thread while the kernel loops for ~3 s): documents the
box-to-box spread of the Gram forward (MEASUREMENTS R6e).   python tools/clock_under_load.py"""
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
b, n, d, c = 8192, 105, 1600, 5
z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
w = torch.randn(b, n, n, generator=g, device=dev)
w = (w + w.transpose(1, 2)).contiguous()
e = ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)
cls = torch.arange(c, device=dev).repeat_interleave(n // c)
y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv, mean, noise = torch.linspace(0.8, 1.4, c, device=dev), torch.zeros(c, device=dev), torch.full((c,), 0.1, device=dev)
cw = torch.full((c,), -1.0 / (c * n), device=dev)


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            j = json.loads(r.stdout)["card0"]
            sclk = [v for k, v in j.items() if "sclk" in k.lower()]
            pw = [v for k, v in j.items() if "power" in k.lower() and "(w)" in k.lower()]
            out.append((float(str(sclk[0]).strip("()Mhz ")) if sclk else float("nan"), float(pw[0]) if pw else float("nan")))
        except Exception as exc:  # noqa: BLE001 -- best effort: a box without rocm-smi still prints the kernel times
            out.append((float("nan"), float("nan")))
            if len(out) == 1:
                print("rocm-smi poll failed:", exc, flush=True)
        time.sleep(0.05)


# the fused front end of the drop-in class (2048 episodes from trunk features), the 20-way band path (1024 episodes of 420 rows) and the QMUL head's small kernels
x = (torch.randn(2048, n, d, generator=g, device=dev).abs() + 1.0)
gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
fe = ops.gram_bn_train(x, gamma, beta)
z4 = torch.nn.functional.normalize(torch.randn(1024, 420, 512, generator=g, device=dev), dim=2).contiguous()
e4 = ops.gram(z4, None, ops.KERNEL_LINEAR_UNIT)
w4 = torch.randn(1024, 420, 420, generator=g, device=dev)
w4 = (w4 + w4.transpose(1, 2)).contiguous()
cls4 = torch.arange(20, device=dev).repeat_interleave(21)
y4 = torch.where(cls4.unsqueeze(0) == torch.arange(20, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv4, mean4, noise4, cw4 = torch.linspace(0.8, 1.4, 20, device=dev), torch.zeros(20, device=dev), torch.full((20,), 0.1, device=dev), torch.full((20,), -1.0 / 8400, device=dev)
z0 = torch.randn(8192, 19, 2916, generator=g, device=dev) * 0.05
w0 = torch.randn(8192, 19, 19, generator=g, device=dev) * 0.1
ls0 = torch.tensor([1.3], device=dev)

for name, fn in (("dkt_gram_f32", lambda: ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)),
                 ("dkt_gram_bwd_f32", lambda: ops.gram_bwd(w, z, None, unit_rows=True, w_symmetric=True)),
                 ("dkt_mll_f32", lambda: ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)),
                 ("gram_bn_train 2048", lambda: ops.gram_bn_train(x, gamma, beta)),
                 ("gram N=420 x1024", lambda: ops.gram(z4, None, ops.KERNEL_LINEAR_UNIT)),
                 ("gram_bwd N=420", lambda: ops.gram_bwd(w4, z4, None, unit_rows=True, w_symmetric=True)),
                 ("mll band N=420", lambda: ops.mll(e4, y4, sv4, mean4, noise4, want_grad=True, cls_weight=cw4)),
                 ("gram 19x2916 rbf", lambda: ops.gram(z0, None, ops.KERNEL_RBF, ls0)),
                 ("gram_bwd 19x2916", lambda: ops.gram_bwd(w0, z0))):
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
    while time.perf_counter() - tw < 2.5:
        for _ in range(10):
            fn()
        reps += 10
        torch.cuda.synchronize()
    t1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = t0.elapsed_time(t1) / reps
    sc = [s for s, _ in samples[2:] if s == s]
    pw = [p for _, p in samples[2:] if p == p]
    print("%-18s %.4f ms per call   sclk under load: median %s MHz (min %s, max %s, %d samples)   socket power median %s W"
          % (name, ms, statistics.median(sc) if sc else "?", min(sc) if sc else "?", max(sc) if sc else "?", len(sc), statistics.median(pw) if pw else "?"), flush=True)
