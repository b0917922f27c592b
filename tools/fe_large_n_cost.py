"""What the un-fused front end costs at the 20-way shapes (N = 420 > 128: bn_out in train mode + F.normalize run as torch ops in front of dkt_gram_f32, and autograd
runs back through them): time of exactly those torch ops, forward + backward, next to the GP part of the step (Gram + marginal likelihood + Gram backward).
Measurement tooling."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for (b, c, n, d, reps) in [(1024, 20, 420, 512, 5), (64, 20, 420, 512, 20), (1, 20, 420, 512, 50)]:
    x = torch.randn(b, n, d, generator=g, device=dev).abs().requires_grad_(True)           # trunk features (post-ReLU / pooling)
    bn = torch.nn.BatchNorm1d(d).to(dev).train()
    per = n // c
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.full((c,), 0.7, device=dev, requires_grad=True)
    mean = torch.zeros(c, device=dev, requires_grad=True)
    noise = torch.full((c,), 0.1, device=dev, requires_grad=True)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    gz = torch.randn(b, n, d, generator=g, device=dev)

    def front_end_only():
        x.grad = None
        z = torch.stack([torch.nn.functional.normalize(bn(x[i]), p=2, dim=1) for i in range(b)]) if b <= 64 else None
        if z is None:       # per-episode statistics for a large batch in one torch call: the features of an episode normalised over its own rows
            m = x.mean(1, keepdim=True)
            v = x.var(1, unbiased=False, keepdim=True)
            z = torch.nn.functional.normalize((x - m) * torch.rsqrt(v + 1e-5) * bn.weight + bn.bias, p=2, dim=2)
        z.backward(gz)

    zc = torch.nn.functional.normalize(x.detach(), dim=2).requires_grad_(True)

    def gp_part():
        zc.grad = None
        obj, *_ = ops.episode_loss_linear(zc, y, sv, mean, noise, cw, unit_rows=True)
        obj.mean().backward()

    gamma, beta = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)

    def whole_step_hip():                                    # what DKT._episode_loss_from_trunk runs since round 4: the streaming front end + the GP part
        x.grad = None
        obj, *_ = ops.episode_loss_bn(x, gamma, beta, y, sv, mean, noise, cw)
        obj.mean().backward()

    t_all = timed(whole_step_hip, reps)
    t_fe, t_gp = timed(front_end_only, reps), timed(gp_part, reps)
    print("B=%d N=%d D=%d: the whole step from trunk features through the HIP front end %.3f ms (torch front end + GP part: %.3f ms)" % (b, n, d, t_all, t_fe + t_gp))
    print("B=%d N=%d D=%d: torch bn_out(train) + F.normalize, forward + backward %.3f ms; Gram + marginal likelihood + Gram backward %.3f ms  (front end = %.1f %% of the two)"
          % (b, n, d, t_fe, t_gp, 100 * t_fe / (t_fe + t_gp)), flush=True)
