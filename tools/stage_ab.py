"""Same-box A/B of the one-round-trip LDS staging of W (and E) in the Gram-backward prologues (round 5, dkt_split.h LdsStage; twins library):
  DKT_LDS_STAGE_OLD=1  the scalar copy loop that hipcc drains per unrolled trip  vs  all loads in flight at once (default)
Each variant is timed in turn, three rounds, so that clock / thermal drift does not favour one; outputs are compared bitwise.
(profiles/r05/v5_stage_ab.log also holds the software-pipelined slab loop of the fused backward that was measured with this tool and dropped.)
    python tools/stage_ab.py"""
import os
import sys

os.environ["DKT_TWINS"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

dkt_amd = importlib.import_module("deep-kernel-transfer_amd")
ops = dkt_amd.ops
dev = torch.device("cuda:0")


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        out = fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps, out


def setenv(d):
    for k in ("DKT_LDS_STAGE_OLD",):
        os.environ.pop(k, None)
    os.environ.update(d)


variants = [("new", {"DKT_LDS_STAGE_OLD": "0"}), ("old_stage", {"DKT_LDS_STAGE_OLD": "1"})]
for (b, n, d) in [(8192, 105, 1600), (8192, 85, 512), (8192, 105, 64), (2048, 105, 1600)]:
    g = torch.Generator(device=dev).manual_seed(n + d)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, device=dev, generator=g), dim=2)
    w = torch.randn(b, n, n, device=dev, generator=g) * 0.01
    w = (w + w.transpose(1, 2)).contiguous()
    res, ref = {}, None
    for rnd in range(3):
        for name, env in variants:
            setenv(env)
            ms, out = timed(lambda: ops.gram_bwd(w, z, None, unit_rows=True))
            res.setdefault(name, []).append(ms)
            if ref is None:
                ref = out
            else:
                assert torch.equal(out, ref), name
    alg = b * (2 * n * d + n * n) * 4
    print("gram_bwd level-1 B=%d N=%d D=%d: " % (b, n, d) + "  ".join("%s %.4f ms (%.3f of 8 TB/s)" % (k, min(v), alg / min(v) / 1e9 / 8.0) for k, v in res.items()) + "  bitwise equal", flush=True)
    del z, w

variants = [("new", {"DKT_LDS_STAGE_OLD": "0"}), ("old_stage", {"DKT_LDS_STAGE_OLD": "1"})]
for (b, n, d) in [(2048, 105, 1600), (8192, 105, 1600), (2048, 85, 512), (2048, 128, 1600), (2048, 80, 640)]:
    g = torch.Generator(device=dev).manual_seed(n + d + 1)
    x = torch.randn(b, n, d, device=dev, generator=g).abs() * 2.0 + 1.0
    gamma, beta = 0.5 + torch.rand(d, device=dev, generator=g), 0.2 * torch.randn(d, device=dev, generator=g)
    setenv({"DKT_LDS_STAGE_OLD": "0"})
    e, rnorm, st = ops.gram_bn_train(x, gamma, beta, 1e-5)
    w = torch.randn(b, n, n, device=dev, generator=g) * 0.01
    w = (w + w.transpose(1, 2)).contiguous()
    gobj = torch.linspace(0.5, 1.5, b, device=dev)
    res, ref = {}, None
    for rnd in range(3):
        for name, env in variants:
            setenv(env)
            ms, out = timed(lambda: ops.gram_bn_bwd(w, e, x, st["a"], st["s"], rnorm, st["mean"], st["rstd"], gobj))
            res.setdefault(name, []).append(ms)
            if ref is None:
                ref = out
            else:
                assert all(torch.equal(o, r) for o, r in zip(out, ref)), name
    alg = b * (2 * n * n + 2 * n * d + 4 * n + 24 * d) * 4
    print("gram_bn_bwd fused B=%d N=%d D=%d: " % (b, n, d) + "  ".join("%s %.4f ms (%.3f)" % (k, min(v), alg / min(v) / 1e9 / 8.0) for k, v in res.items()) + "  bitwise equal", flush=True)
    del x, e, w
