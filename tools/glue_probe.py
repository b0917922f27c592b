"""What a step costs beyond its product kernels: the drop-in step from trunk features (cfg2 shape, 2048 / 8192 episodes) and the feature-space step (cfg1, 8192),
timed with and without the per-kernel HIP events bench.py records, for rocprofv3 --kernel-trace --stats (every launch of the step by name).  Measurement tooling.

    python tools/glue_probe.py [trunk|cfg1|all]
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/glue -- python tools/glue_probe.py trunk
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dkt_amd import ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda", 0)


def timed(fn, steps, events):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ops.kernel_timing(events)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kt = {k: round(v[1], 4) for k, v in ops.kernel_timing_results().items()} if events else {}
    ops.kernel_timing(False)
    return 1e3 * dt, kt


def trunk_step(b, n=105, d=1600, c=5, per=21):
    g = torch.Generator(device=dev).manual_seed(4321)
    x = (torch.randn(b, n, d, generator=g, device=dev).abs() + 1.0).requires_grad_(True)
    gamma = torch.ones(d, device=dev, requires_grad=True)
    beta = torch.zeros(d, device=dev, requires_grad=True)
    raw_s, mean = bench.perturbed_hypers(c, 99, dev)
    raw_s.requires_grad_(True)
    mean.requires_grad_(True)
    noise = torch.full((c,), 0.1, device=dev)
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    leaves = (x, gamma, beta, raw_s, mean)

    def fn():
        for t in leaves:
            t.grad = None
        outs = ops.episode_loss_bn(x, gamma, beta, y, torch.nn.functional.softplus(raw_s), mean, noise, cw)
        outs[0].mean().backward()
    return fn


if what in ("trunk", "all"):
    for b in (2048, 8192):
        fn = trunk_step(b)
        a, _ = timed(fn, 20, False)
        e, kt = timed(fn, 20, True)
        a2, _ = timed(fn, 20, False)
        print("from trunk features, %d episodes: %.4f / %.4f ms per step without events, %.4f with; kernels %s (sum %.4f)" % (b, a, a2, e, kt, sum(kt.values())), flush=True)
        del fn
        torch.cuda.empty_cache()

if what in ("cfg1", "all"):
    for cfg, b in (("cfg1", 8192), ("cfg2", 8192)):
        step, _ = bench._workload(cfg, b, dev, 0, True)
        a, _ = timed(step, 20, False)
        e, kt = timed(step, 20, True)
        a2, _ = timed(step, 20, False)
        print("%s, %d episodes: %.4f / %.4f ms per step without events, %.4f with; kernels %s (sum %.4f)" % (cfg, b, a, a2, e, kt, sum(kt.values())), flush=True)
        del step
        torch.cuda.empty_cache()

if what in ("cfg4",):
    r = bench._aux_paths(dev, "cfg4", 512, 64)["from_trunk_features"]
    print("cfg4 from trunk features, 512 episodes: %.4f ms per step; kernels %s" % (r["ms_per_step"], r["kernels_ms"]), flush=True)
    print("   hbm_frac %s" % {k: v["frac"] for k, v in r["roofline"].items()}, flush=True)

if what in ("cfg1_20way", "lowrank"):
    for cfg, b in (("cfg1", 8192), ("cfg1_20way", 2048)):
        step, _ = bench._workload(cfg, b, dev, 0, True)
        a, _ = timed(step, 20, False)
        e, kt = timed(step, 20, True)
        a2, _ = timed(step, 20, False)
        print("%s, %d episodes: %.4f / %.4f ms per step without events, %.4f with; kernels %s (sum %.4f)" % (cfg, b, a, a2, e, kt, sum(kt.values())), flush=True)
        del step
        torch.cuda.empty_cache()
