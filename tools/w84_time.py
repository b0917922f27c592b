import os as _os; _os.environ.setdefault("DKT_TWINS", "1")   # the variant switches this tool flips live in libdkt_twins.so (ops._lib_now)
import os, sys, torch
sys.path.insert(0, os.getcwd())
import dkt_amd
from dkt_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
for (b, c, n) in [(1024, 20, 420), (1024, 20, 380)]:
    per = n // c
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.full((c,), 0.7, device=dev); mean = torch.zeros(c, device=dev); noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    z = torch.nn.functional.normalize(torch.randn(b, n, 128, generator=g, device=dev), dim=2).contiguous()
    e = ops.gram(z)
    for env in ({"DKT_MLL_TILED_WNW": "4"}, {"DKT_MLL_TILED_WNW": "84"}, {"DKT_MLL_TILED_WNW": "4"}, {"DKT_MLL_TILED_WNW": "84"}):
        os.environ.update(env)
        for _ in range(2): ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): o = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
        t.record(); torch.cuda.synchronize()
        print("N=%d %s: %.3f ms  (w checksum %.6e)" % (n, env, s.elapsed_time(t) / 3, o["w"].double().abs().sum().item()), flush=True)
