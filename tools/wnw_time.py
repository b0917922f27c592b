"""W kernel with 8 waves per workgroup: variants of the staged group size (a -DDKT_WRES_NST8=k build) against the default library.  Measurement tooling."""
import os as _os; _os.environ.setdefault("DKT_TWINS", "1")   # the variant switches this tool flips live in libdkt_twins.so (ops._lib_now)
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    import dkt_amd
    for k in sys.argv[2:]:
        os.environ["DKT_EXTRA_HIPCC_FLAGS"] = "-DDKT_WRES_NST8=" + k
        dkt_amd._lib.build(out=os.path.join(ROOT, "deep-kernel-transfer_amd", "libdkt_nst%s.so" % k))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import torch
    import dkt_amd
    from dkt_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    for (b, c, n) in [(1024, 20, 420), (1024, 20, 320), (1024, 20, 380)]:
        per = n // c
        cls = torch.arange(c, device=dev).repeat_interleave(per)
        y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
        sv = torch.full((c,), 0.7, device=dev); mean = torch.zeros(c, device=dev); noise = torch.full((c,), 0.1, device=dev)
        cw = torch.full((c,), -1.0 / (c * n), device=dev)
        z = torch.nn.functional.normalize(torch.randn(b, n, 128, generator=g, device=dev), dim=2).contiguous()
        e = ops.gram(z)
        for env in ({"DKT_MLL_TILED_WNW": "4"}, {"DKT_MLL_TILED_WNW": "8"}, {"DKT_MLL_TILED_WNW": "4"}, {"DKT_MLL_TILED_WNW": "8"}):
            os.environ.update(env)
            for _ in range(2): ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
            torch.cuda.synchronize()
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3): o = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
            t.record(); torch.cuda.synchronize()
            print("  %s N=%d %s: %.3f ms  (w checksum %.6e)" % (sys.argv[2], n, env, s.elapsed_time(t) / 3, o["w"].double().abs().sum().item()), flush=True)
    sys.exit(0)
for k in ["default"] + sys.argv[1:]:
    env = dict(os.environ)
    if k != "default":
        env["DKT_AMD_LIB"] = os.path.join(ROOT, "deep-kernel-transfer_amd", "libdkt_nst%s.so" % k)
    subprocess.run([sys.executable, os.path.abspath(__file__), "one", k], env=env, timeout=600)
