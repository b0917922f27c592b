"""Feasibility probe: the shared-reduction (band) path of dkt_mll_f32 on the 64 x 64 models of the feature-space episode (DKT_MLL_FORCE_BAND below N = 128) against the
default wave-per-episode kernel and float64.  NEEDS a library in which dkt_mll_band_supports() admits N >= 48 under DKT_MLL_FORCE_BAND and dkt_mll_f32 tries that path before
the N <= 127 kernels (a three-line change, not shipped).  Result (MEASUREMENTS R6d): correct (1e-6) and 2.7 x SLOWER than the wave-per-episode kernel.
python tools/experiments/band_small_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


for (b, c, n, dp) in [(2048, 20, 420, 64), (8192, 5, 105, 64), (512, 20, 420, 64), (2048, 20, 420, 48), (2048, 12, 240, 96)]:
    g = torch.Generator(device=dev).manual_seed(n + c)
    z = torch.nn.functional.normalize(torch.randn(b, n, dp, generator=g, device=dev), dim=2)
    a = (z.transpose(1, 2) @ z).contiguous()                                  # [B, dp, dp]
    cls = torch.arange(c, device=dev).repeat_interleave(n // c)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0)
    p = (y.unsqueeze(0) @ z).contiguous()                                      # [B, C, dp] per-episode targets
    sv = torch.linspace(0.7, 1.5, c, device=dev)
    mean = torch.zeros(c, device=dev)
    noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    f_def = lambda: ops.mll(a, p, sv, mean, noise, want_grad=True, cls_weight=cw, no_kappa_guard=True)
    f_band = lambda: ops.mll(a, p, sv, mean, noise, want_grad=True, cls_weight=cw, force_band=True)
    od, ob = f_def(), f_band()
    # float64 on 64 episodes
    k = 64
    a64, p64 = a[:k].double(), p[:k].double()
    eye = torch.eye(dp, device=dev, dtype=torch.float64)
    kc = sv.double().view(1, c, 1, 1) * a64.unsqueeze(1) + noise.double().view(1, c, 1, 1) * eye
    ch = torch.linalg.cholesky(kc)
    al = torch.cholesky_solve(p64.unsqueeze(-1), ch).squeeze(-1)
    logdet = 2 * torch.log(torch.diagonal(ch, dim1=-2, dim2=-1)).sum(-1)
    logp = -0.5 * ((p64 * al).sum(-1) + logdet + dp * 1.8378770664093453)
    kinv = torch.cholesky_inverse(ch)
    w = (0.5 * cw.double().view(1, c, 1, 1) * sv.double().view(1, c, 1, 1) * (al.unsqueeze(-1) * al.unsqueeze(-2) - kinv)).sum(1)
    line = "B=%d C=%d N'=%d:" % (b, c, dp)
    for name, o in (("default", od), ("band", ob)):
        el = ((o["logp"][:k].double() - logp).abs() / logp.abs()).max().item()
        ea = ((o["alpha"][:k].double() - al).norm() / al.norm()).item()
        ew = ((o["w"][:k].double() - w).norm() / w.norm()).item()
        line += "  %s: logp %.1e alpha %.1e W %.1e info %d" % (name, el, ea, ew, int(o["info"].abs().max().item()))
    line += "  | default %.4f ms  band %.4f ms" % (timed(f_def), timed(f_band))
    print(line, flush=True)
