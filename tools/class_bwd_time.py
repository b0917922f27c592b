import os as _os; _os.environ.setdefault("DKT_TWINS", "1")   # the variant switches this tool flips live in libdkt_twins.so (ops._lib_now)
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import dkt_amd
from dkt_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
for (b, c, n) in [(2048, 5, 105), (2048, 5, 85), (64, 20, 420)]:
    base = torch.rand(b, n, n, generator=g, device=dev) * 3
    w = torch.randn(b, c, n, n, generator=g, device=dev)
    w = (w + w.transpose(2, 3)).contiguous()
    p = torch.linspace(1.0, 2.0, c, device=dev)
    for env in ({}, {"DKT_CLASS_BWD_V4": "0"}, {}):
        os.environ.update(env)
        for _ in range(3): ops.class_kernel_bwd(w, base, 0, 0, p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ops.class_kernel_bwd(w, base, 0, 0, p)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 100
        for k in env: os.environ.pop(k)
        print("class_kernel_bwd B=%d C=%d N=%d %s: %.3f ms (%.2f TB/s of W)" % (b, c, n, "dword kernel" if env else "16-byte kernel", ms, w.numel() * 4 / ms / 1e9), flush=True)
