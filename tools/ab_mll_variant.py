"""A/B of a variant build of the library (DKT_AMD_LIB) against the default one on the marginal-likelihood training call: bitwise
comparison of every output and time per launch, alternating.  usage: python tools/ab_mll_variant.py <variant .so> [N ...]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import dkt_amd  # noqa: F401
    from dkt_amd import ops
    dev = torch.device("cuda", 0)
    out_path = sys.argv[2]
    res = {}
    for n in [int(v) for v in sys.argv[3:]]:
        c, d, b = 5, 64, 8192
        g = torch.Generator(device=dev).manual_seed(n)
        z = torch.nn.functional.normalize(torch.randn(2048, n, d, generator=g, device=dev), dim=2)
        e = ops.gram(z).repeat(4, 1, 1).contiguous()
        cls = torch.arange(c, device=dev).repeat_interleave(n // c)
        y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
        sv = torch.full((c,), 0.69, device=dev) * (1.0 + 0.03 * torch.arange(c, device=dev))
        mean, noise = torch.zeros(c, device=dev), torch.full((c,), 0.1, device=dev)
        cw = torch.full((c,), -1.0 / (c * n), device=dev)
        for _ in range(3):
            o = ops.mll(e, y, sv, mean, noise, cls_weight=cw, want_grad=True)
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30):
            o = ops.mll(e, y, sv, mean, noise, cls_weight=cw, want_grad=True)
        t.record()
        torch.cuda.synchronize()
        res[n] = dict(ms=s.elapsed_time(t) / 30, **{k: o[k].cpu() for k in ("logp", "alpha", "w", "dsv", "dmean", "dnoise")})
    torch.save(res, out_path)
    sys.exit(0)

variant = os.path.abspath(sys.argv[1])
ns = sys.argv[2:] or ["105", "85"]
import torch  # noqa: E402
outs = {}
for rnd in range(2):
    for name, lib in (("default", None), ("variant", variant)):
        env = dict(os.environ)
        if lib:
            env["DKT_AMD_LIB"] = lib
        path = "/tmp/ab_%s_%d.pt" % (name, rnd)
        subprocess.run([sys.executable, __file__, "--child", path] + ns, env=env, check=True)
        outs.setdefault(name, []).append(torch.load(path))
for n in [int(v) for v in ns]:
    a, b = outs["default"], outs["variant"]
    same = all(torch.equal(a[0][n][k], b[0][n][k]) for k in ("logp", "alpha", "w", "dsv", "dmean", "dnoise"))
    print("N=%d: default %.4f / %.4f ms, variant %.4f / %.4f ms, outputs bitwise equal: %s" % (n, a[0][n]["ms"], a[1][n]["ms"], b[0][n]["ms"], b[1][n]["ms"], same))
