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


for name, fn in (("dkt_gram_f32", lambda: ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)),
                 ("dkt_gram_bwd_f32", lambda: ops.gram_bwd(w, z, None, unit_rows=True, w_symmetric=True)),
                 ("dkt_mll_f32", lambda: ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw))):
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
    while time.perf_counter() - tw < 3.0:
        for _ in range(50):
            fn()
        reps += 50
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
