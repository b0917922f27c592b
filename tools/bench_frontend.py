"""Training episode from the trunk output X (before bn_out): fused HIP front end (dkt_bn_stats_f32 + dkt_gram_bn_f32 +
dkt_mll_f32 + dkt_gram_bn_bwd_f32) vs torch BatchNorm/normalize in front of the same hot path.  cfg2 shape."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
c, per, d = 5, 21, 1600
n = c * per
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.randn(b, n, d, generator=g, device=dev).abs() + 1.0).requires_grad_(True)
gamma = torch.ones(d, device=dev, requires_grad=True)
beta = torch.zeros(d, device=dev, requires_grad=True)
raw_s = torch.zeros(c, device=dev, requires_grad=True)
mean = torch.zeros(c, device=dev, requires_grad=True)
noise = torch.full((c,), 0.1, device=dev)
cls = torch.arange(c, device=dev).repeat_interleave(per)
y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
cw = torch.full((c,), -1.0 / (c * n), device=dev)


def fused_two_pass():
    os.environ["DKT_FUSED_STATS"] = "0"
    try:
        return fused()
    finally:
        os.environ["DKT_FUSED_STATS"] = "1"


def fused():
    for t in (x, gamma, beta, raw_s, mean):
        t.grad = None
    obj = ops.episode_loss_bn(x, gamma, beta, y, torch.nn.functional.softplus(raw_s), mean, noise, cw)[0]
    obj.mean().backward()
    return obj


def unfused():
    for t in (x, gamma, beta, raw_s, mean):
        t.grad = None
    mu = x.mean(1, keepdim=True)
    var = x.var(1, unbiased=False, keepdim=True)
    z = (x - mu) * torch.rsqrt(var + 1e-5) * gamma + beta
    z = torch.nn.functional.normalize(z, p=2, dim=2)
    obj = ops.episode_loss_linear(z, y, torch.nn.functional.softplus(raw_s), mean, noise, cw)[0]
    obj.mean().backward()
    return obj


for name, fn in (("fused HIP front end", fused), ("fused, stats in a separate pass", fused_two_pass), ("torch BN + normalize", unfused)):
    for _ in range(3):
        o = fn()
    torch.cuda.synchronize()
    ops.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(10):
        o = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    kt = ops.kernel_timing_results()
    ops.kernel_timing(False)
    print("%-32s %.3f ms per %d episodes = %.0f episodes/s   kernels: %s" % (name, 1e3 * dt, b, b / dt,
          {k: round(v[1], 4) for k, v in kt.items()}), flush=True)
    if name == "fused HIP front end":
        ref_x = x.grad.clone()
        ref_o = o.detach().clone()
print("max |obj fused - unfused| = %.3e   rel-L2 dX = %.3e" % ((ref_o - o.detach()).abs().max().item(),
      ((ref_x - x.grad).norm() / x.grad.norm()).item()))

# which of the two fp32 gradients is right?  float64 autograd of the same formulation on the first episodes
unf_x = x.grad[:2].clone()
x64 = x.detach()[:2].double().cpu().requires_grad_(True)
mu = x64.mean(1, keepdim=True)
var = x64.var(1, unbiased=False, keepdim=True)
z = (x64 - mu) * torch.rsqrt(var + 1e-5)
z = torch.nn.functional.normalize(z, p=2, dim=2)
e = z @ z.transpose(1, 2)
sv = torch.nn.functional.softplus(torch.zeros(c, dtype=torch.float64))
tot = 0.0
for i in range(2):
    for cc in range(c):
        k = sv[cc] * e[i] + 0.1 * torch.eye(n, dtype=torch.float64)
        L = torch.linalg.cholesky(k)
        r = y[cc].double().cpu()
        al = torch.cholesky_solve(r.unsqueeze(1), L).squeeze(1)
        logp = -0.5 * (r @ al) - torch.log(torch.diagonal(L)).sum() - 0.5 * n * 1.8378770664093453
        tot = tot + (-1.0 / (c * n)) * logp / b
tot.backward()
ref = x64.grad
print("rel-L2 vs float64 autograd:  fused %.3e   torch-fp32 front end %.3e" % (
    ((ref_x[:2].double().cpu() - ref).norm() / ref.norm()).item(), ((unf_x.double().cpu() - ref).norm() / ref.norm()).item()))
