"""From trunk features at D <= 64 < N <= 128 (Conv4S / Omniglot behind bn_out): the fused N x N front end (dkt_gram_bn_train_f32 -> dkt_mll_f32 -> dkt_gram_bn_bwd_f32)
against the streaming front end + the feature-space episode (dkt_bn_stats_f32 -> dkt_affine_normalize_f32 -> dkt_lowrank_* -> dkt_normalize_bn_bwd_f32), same box.
    python tools/fe_lowrank_ab.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("deep-kernel-transfer_amd").ops
dev = torch.device("cuda:0")
for (b, c, per, d) in [(8192, 5, 21, 64), (4096, 5, 21, 64), (2048, 5, 21, 64), (1024, 5, 21, 64), (8192, 5, 16, 64), (8192, 5, 17, 64)]:
    n = c * per
    g = torch.Generator(device=dev).manual_seed(n + d)
    x = (torch.randn(b, n, d, generator=g, device=dev).abs() + 1.0).requires_grad_(True)
    gamma = torch.ones(d, device=dev, requires_grad=True)
    beta = torch.zeros(d, device=dev, requires_grad=True)
    raw_s = (0.1 * torch.randn(c, device=dev, generator=g)).requires_grad_(True)
    mean = (0.05 * torch.randn(c, device=dev, generator=g)).requires_grad_(True)
    noise = torch.full((c,), 0.1, device=dev)
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    cw = torch.full((c,), -1.0 / (c * n), device=dev)

    def step():
        for t in (x, gamma, beta, raw_s, mean):
            t.grad = None
        outs = ops.episode_loss_bn(x, gamma, beta, y, torch.nn.functional.softplus(raw_s), mean, noise, cw)
        outs[0].sum().backward()
        return outs[0].detach(), x.grad

    res, outs = {}, {}
    for rnd in range(3):
        for mode in ("1", "force"):
            os.environ["DKT_LOWRANK"] = mode
            for _ in range(2):
                step()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(5):
                o = step()
            t1.record()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(t0.elapsed_time(t1) / 5)
            outs[mode] = (o[0].clone(), o[1].clone())
    del os.environ["DKT_LOWRANK"]
    dobj = ((outs["1"][0] - outs["force"][0]).abs().max() / outs["1"][0].abs().max()).item()
    ddx = ((outs["1"][1] - outs["force"][1]).norm() / outs["1"][1].norm()).item()
    print("B=%d N=%d D=%d: fused N x N front end %.4f ms (%.2f M eps/s)   streaming front end + feature space %.4f ms (%.2f M eps/s)   obj rel diff %.1e, dX rel diff %.1e"
          % (b, n, d, min(res["1"]), b / min(res["1"]) / 1e3, min(res["force"]), b / min(res["force"]) / 1e3, dobj, ddx), flush=True)
