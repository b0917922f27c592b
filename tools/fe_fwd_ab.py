"""Same-box A/B of the train-mode fused forward (dkt_gram_bn_train_f32): DKT_GRAM_BN_F16 = 0 (3-way bf16 split, the round-3 kernel), 1 (the pipelined
scaled 2-way f16 kernel of round 5, default).  Prints ms per call, the HBM fraction on the algorithmic bytes
(X once + E + the statistics) and the worst error against float64 on a few episodes.   python tools/fe_fwd_ab.py [B N D]..."""
import os
import sys

os.environ["DKT_TWINS"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

dkt_amd = importlib.import_module("deep-kernel-transfer_amd")
ops = dkt_amd.ops

shapes = [(2048, 105, 1600), (8192, 105, 1600), (2048, 85, 512), (2048, 128, 1600), (2048, 80, 1600), (2048, 50, 640)]
if len(sys.argv) > 3:
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
dev = torch.device("cuda:0")
for (b, n, d) in shapes:
    g = torch.Generator(device=dev).manual_seed(n + d)
    x = (torch.randn(b, n, d, device=dev, generator=g).abs() * (0.2 + 2.8 * torch.rand(1, 1, d, device=dev, generator=g)) + 5.0 * torch.rand(1, 1, d, device=dev, generator=g))
    gamma = 0.5 + torch.rand(d, device=dev, generator=g)
    beta = 0.2 * torch.randn(d, device=dev, generator=g)
    x64 = x[:3].double().cpu()
    y = (x64 - x64.mean(1, keepdim=True)) / torch.sqrt(x64.var(1, unbiased=False, keepdim=True) + 1e-5) * gamma.double().cpu() + beta.double().cpu()
    zn = torch.nn.functional.normalize(y, dim=2)
    ref = zn @ zn.transpose(1, 2)
    alg = b * (n * d * 4 + n * n * 4 + 4 * n + 20 * d)
    for v in ("0", "1"):
        os.environ["DKT_GRAM_BN_F16"] = v
        e, rn, st = ops.gram_bn_train(x, gamma, beta, 1e-5)
        torch.cuda.synchronize()
        flagged = int(torch.isnan(rn[:, 0]).sum().item())
        err = (e[:3].double().cpu() - ref).abs().max().item()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ops.gram_bn_train(x, gamma, beta, 1e-5)
        t0.record()
        reps = 20
        for _ in range(reps):
            ops.gram_bn_train(x, gamma, beta, 1e-5)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / reps
        print("B=%d N=%d D=%d  F16=%s  %.4f ms  %.2f TB/s  frac %.3f  max|E - float64| %.2e  nan-flags left %d" % (b, n, d, v, ms, alg / ms / 1e9, alg / ms / 1e9 / 8.0, err, flagged), flush=True)
    del os.environ["DKT_GRAM_BN_F16"]
