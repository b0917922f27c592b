#!/usr/bin/env python
"""bench.py -- episodes/sec of the DKT hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic episodes per GPU: training episode
forward + backward, i.e. Gram build (dkt_gram_f32) -> C jittered Choleskys / log-dets / solves / MLL and
their gradient pieces (dkt_mll_f32) -> dZ = (W + W^T) Z (dkt_gram_bwd_f32), plus the [B]-sized torch
glue.  Workload = BASELINE.json configs[1]-class headline "5-way 5-shot Conv4" shape the metric is quoted
on (N=105, D=1600, C=5); inputs (normalised features Z) are resident in HBM before the
timed region.  Episodes shard across ranks with no data-path collective (weak scaling); the only exchange
is the all-reduce of the 2C shared GP hyper-parameter gradients per step.

  python bench.py [--gpus N --steps K --warmup W --episodes B --config cfg2]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32-input MFMA = fp32 vector peak

CONFIGS = {   # name -> (n_way, n_support, n_query, D, description)
    "cfg1": (5, 5, 16, 64, "Omniglot 5-way 5-shot, Conv4S features"),
    "cfg2": (5, 5, 16, 1600, "CUB 5-way 5-shot, Conv4 features (headline)"),
    "cfg3": (5, 1, 16, 512, "miniImagenet 5-way 1-shot, ResNet10 features"),
    # 20-way: N = 420 is what train_loop builds (20 x (5 + 16)); BASELINE.json quotes a 320 x 320 Gram
    "cfg4": (20, 5, 16, 512, "miniImagenet 20-way 5-shot, ResNet18 features (N = 420: blocked large-N MLL path)"),
    "cfg4_n320": (20, 1, 15, 512, "20-way, N = 320 (the Gram size BASELINE.json quotes)"),
}


def synthetic_batch(b, n, d, seed, device):
    """Zraw ~ N(0,1) -> BN1d(train, gamma 1, beta 0) -> L2 normalise, made on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    z = torch.randn(b, n, d, generator=g, device=device, dtype=torch.float32)
    z = (z - z.mean(1, keepdim=True)) / torch.sqrt(z.var(1, unbiased=False, keepdim=True) + 1e-5)
    return torch.nn.functional.normalize(z, p=2, dim=2).contiguous()


def perturbed_hypers(c, seed, device):
    g = torch.Generator().manual_seed(seed)
    raw_s = (torch.randn(c, generator=g) * 0.5).to(device)
    mean = (torch.randn(c, generator=g) * 0.1).to(device)
    return raw_s, mean


def cpu_baseline(z_cpu, n_way, raw_s, mean, budget_s=12.0):
    """The oracle's fp32 GPyTorch-structured port (per-class loop, dense Cholesky, autograd backward),
    B=1 sequential episodes as the reference runs them, timed on this box's host cores."""
    from oracle import dkt_oracle_torch as T
    res = {}
    ncores = os.cpu_count() or 1
    for threads in sorted({1, min(ncores, 8)}):
        torch.set_num_threads(threads)
        for i in range(3):     # warm-up
            zi = z_cpu[i % z_cpu.shape[0]].clone().requires_grad_(True)
            T.cpu_baseline_train_episode(zi, n_way, raw_s.clone().requires_grad_(True), mean.clone().requires_grad_(True))
        n_done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            zi = z_cpu[n_done % z_cpu.shape[0]].clone().requires_grad_(True)
            T.cpu_baseline_train_episode(zi, n_way, raw_s.clone().requires_grad_(True), mean.clone().requires_grad_(True))
            n_done += 1
        res[threads] = n_done / (time.perf_counter() - t0)
    best = max(res, key=res.get)
    return res, best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--episodes", type=int, default=None,
                    help="episodes per step per GPU (SURVEY.md 8d: B in {1, 64, 1024, 8192}); default 8192, 1024 for the 20-way shapes")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-test-time", action="store_true",
                    help="skip the separately reported forward-only test-time episode (profiling runs: keeps the per-kernel "
                         "averages of the trace to the training step's launches)")
    args = ap.parse_args()

    import dkt_amd
    from dkt_amd import ops, distributed

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = distributed.init_from_env("nccl") if world > 1 else 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the DKT hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dkt_amd._lib.needs_build() and rank == 0:
        dkt_amd._lib.build()
    if world > 1:
        torch.distributed.barrier()

    c, s, q, d, desc = CONFIGS[args.config]
    n = c * (s + q)
    b = args.episodes or (8192 if n <= 128 else 1024)
    z = synthetic_batch(b, n, d, 1234 + rank, dev).requires_grad_(True)
    raw_s, mean = perturbed_hypers(c, 99, dev)
    raw_s.requires_grad_(True)
    mean.requires_grad_(True)
    noise = torch.full((c,), 0.1, device=dev)
    cls = torch.arange(c, device=dev).repeat_interleave(s + q)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    bucket = distributed.GradBucket([raw_s, mean])

    UNIT_ROWS = os.environ.get("DKT_BENCH_UNIT", "1") != "0"

    def step():
        z.grad = None
        raw_s.grad = None
        mean.grad = None
        sv = torch.nn.functional.softplus(raw_s)
        # the features are bn_out'ed + L2-normalised rows (the cossim / bncossim contract): unit_rows lets the Gram kernels
        # take the scaled 2-way f16 split; DKT_BENCH_UNIT=0 runs the range-agnostic 3-way bf16 split instead
        obj, logp, alpha, info, jit, e = ops.episode_loss_linear(z, y, sv, mean, noise, cw, unit_rows=UNIT_ROWS)
        loss = obj.mean()
        loss.backward()
        bucket.allreduce_mean()          # the path's only exchange: shared hyper-parameter gradients
        return loss, logp, info

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    ops.kernel_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, logp, info = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    ktimes = ops.kernel_timing_results()
    ops.kernel_timing(False)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ok = int(info.abs().max().item()) == 0 and bool(torch.isfinite(loss))
    # same inputs -> bitwise the same outputs (no atomics, fixed reduction orders): a hand-off race in a kernel would
    # show up here as a handful of differing episodes
    _, logp_a, _ = step()
    grad_a = z.grad.clone()
    _, logp_b, _ = step()
    deterministic = bool(torch.equal(logp_a, logp_b)) and bool(torch.equal(grad_a, z.grad))
    ok = ok and deterministic

    if rank == 0:
        eps = world * b * args.steps / dt
        # algorithmic bytes / flops per EPISODE, per kernel
        alg = {
            "dkt_gram_f32": dict(bytes=4 * (n * d + n * n), flops=2 * n * n * d),
            "dkt_mll_f32": dict(bytes=4 * (2 * n * n + 3 * c * n), flops=c * (n ** 3 // 3 + 2 * n * n + n ** 3)),
            "dkt_gram_bwd_f32": dict(bytes=4 * (n * n + 2 * n * d), flops=2 * n * n * d),
        }
        kernels = {}
        for name, (cnt, ms) in ktimes.items():
            a = alg[name]
            kernels[name] = dict(launches=cnt, ms=round(ms, 4), gbs=round(a["bytes"] * b / ms / 1e6, 1),
                                 tflops=round(a["flops"] * b / ms / 1e9, 2))
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (bench.py cannot
        # profile itself); linear in the episode count, so scaled when --episodes differs from the profiled run.
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except (OSError, ValueError):
            tj = None

        def roof(name):
            """Both roofs for one kernel; `bound` = the roof it sits closer to (the one that binds)."""
            k = kernels[name]
            traffic, src = None, None
            if tj and args.config == "cfg2" and name in tj.get("kernels", {}):
                traffic = round(tj["kernels"][name]["hbm_bytes"] * b / tj["episodes_per_launch"])
                src = tj["source"]
            f_hbm = k["gbs"] / HBM_PEAK_GBS
            f_f32 = k["tflops"] / MFMA_F32_PEAK_TFLOPS
            hbm = dict(bound="hbm", achieved=k["gbs"], peak=HBM_PEAK_GBS, unit="GB/s", frac=round(f_hbm, 4))
            mat = dict(bound="mfma", achieved=k["tflops"], peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(f_f32, 4))
            # the split Gram kernels run fp32-equivalent flops on the bf16 pipe (6 bf16 MFMAs per fp32 product), so
            # their fp32-equivalent rate may exceed the fp32 matrix peak: HBM is the roof that binds them
            first, other = (hbm, mat) if (name != "dkt_mll_f32" or f_hbm >= f_f32) else (mat, hbm)
            r = dict(kernel=name, **first, traffic=traffic, traffic_unit="bytes/launch (PMC: FETCH_SIZE + WRITE_SIZE at L2<->fabric)",
                     traffic_source=src, algorithmic_bytes_per_launch=alg[name]["bytes"] * b,
                     algorithmic_flops_per_launch=alg[name]["flops"] * b, other_roof=other,
                     avg_launch_ms=k["ms"], episodes_per_launch=b)
            if name == "dkt_mll_f32":
                r["note"] = ("N sequential pivot steps per class (one barrier + one LDS round trip each): bound by VALU issue and "
                             "step latency, not by HBM or MFMA; see DESIGN.md section 4.2")
            return r

        dom = max(kernels, key=lambda k: kernels[k]["ms"]) if kernels else None
        roofline = roof(dom) if dom else None
        roofline_all = {name: roof(name) for name in kernels}
        out = {
            "metric": "episodes/sec", "value": round(eps, 1), "unit": "episodes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s; N=%d D=%d C=%d; training episode fwd+bwd (Gram + %d jittered Cholesky/"
                                   "solve/logdet + MLL + backward)" % (args.config, desc, n, d, c, c),
                       "episodes_per_step_per_gpu": b, "kernel": "bncossim", "parallelism": "episode-dp%d" % world,
                       "arithmetic": ("fp32 results; the two Gram contractions split every fp32 operand into two f16 pieces after an "
                                      "exact power-of-two scaling (unit-norm rows; 22 of 24 significand bits, 3 "
                                      "v_mfma_f32_16x16x32_f16 products, fp32 accumulate), the factorisations in fp32" if UNIT_ROWS else
                                      "fp32 results; the two Gram contractions run as an exact 3-way bf16 split of every fp32 "
                                      "operand (6 v_mfma_f32_16x16x32_bf16 products, fp32 accumulate), the factorisations in fp32")},
            "valid": ok, "deterministic": deterministic, "roofline": roofline, "roofline_gram_build": roofline_all.get("dkt_gram_f32"),
            "roofline_by_kernel": roofline_all, "kernels": kernels,
        }
        if world == 1 and not args.no_test_time:
            # the forward-only test-time episode (`correct`, DKT.py:199-272) reported separately:
            # condition on the 25 support features, predict the 75 queries (Gram, MLL without gradients, cross Gram, mean + arg-max)
            bt = min(b, 4096)
            z_te = synthetic_batch(bt, n, d, 77, dev)        # [support; query] features of a test episode, support rows first
            ns = c * s
            cls_s = torch.arange(c, device=dev).repeat_interleave(s)
            ys = torch.where(cls_s.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
            svt = torch.nn.functional.softplus(raw_s.detach())

            def test_episode():
                # ONE pass over the episode's features: the symmetric episode-resident Gram of [support; query] holds both
                # k(support, support) and k(query, support) (the query-query block is the price of streaming at full rate)
                e_all = ops.gram(z_te, None, ops.KERNEL_LINEAR_UNIT if UNIT_ROWS else ops.KERNEL_LINEAR)
                o = ops.mll(e_all[:, :ns, :ns].contiguous(), ys, svt, mean.detach(), noise)
                return ops.predict(e_all[:, ns:, :ns].contiguous(), o["alpha"], svt, mean.detach())

            for _ in range(3):
                test_episode()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                mu_t, lab_t = test_episode()
            torch.cuda.synchronize()
            dt_t = (time.perf_counter() - t1) / 10
            out["test_time_forward"] = {"value": round(bt / dt_t, 1), "unit": "episodes/s", "episodes_per_step": bt,
                                        "ms_per_step": round(1e3 * dt_t, 4),
                                        "workload": "N_support=%d, N_query=%d, D=%d, C=%d: Gram + MLL (no grad) + cross Gram + posterior mean/arg-max"
                                                    % (c * s, c * q, d, c)}
        if world == 1 and not args.no_cpu_baseline:
            import numpy as np
            from oracle import dkt_oracle as O
            nchk = 4
            zc = z[:nchk].detach().cpu()
            sv64 = torch.nn.functional.softplus(raw_s.detach().cpu().double()).numpy()
            hyp = O.GPHypers(sv64, mean.detach().cpu().double().numpy(), np.full(c, 0.1))
            rel = 0.0
            for i in range(nchk):
                ref = O.train_episode(zc[i].double().numpy(), c, hyp)
                rel = max(rel, float(np.abs((logp[i].cpu().numpy() - ref["logp"]) / ref["logp"]).max()))
            out["mll_rel_err"] = rel
            res, best = cpu_baseline(z[:16].detach().cpu(), c, raw_s.detach().cpu(), mean.detach().cpu())
            out["cpu_baseline"] = {"value": round(res[best], 2), "unit": "episodes/s", "cores": best, "kind": "port",
                                   "sample": "fp32 torch-CPU GPyTorch-structured port of the same training episode "
                                             "(per-class loop, Cholesky, autograd backward), B=1 sequential, ~12 s per "
                                             "thread setting on 16 episodes of the same synthetic Z",
                                   "by_threads": {str(k): round(v, 2) for k, v in res.items()},
                                   "host_cpus": os.cpu_count()}
            out["speedup_vs_cpu"] = round(eps / res[best], 1)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
