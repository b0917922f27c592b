#!/usr/bin/env python
"""bench.py -- episodes/sec of the DKT hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic episodes per GPU: training episode
forward + backward, i.e. Gram build (dkt_gram_f32) -> C jittered Choleskys / log-dets / solves / MLL and
their gradient pieces (dkt_mll_f32) -> dZ = (W + W^T) Z (dkt_gram_bwd_f32), plus the [B]-sized torch
glue, plus the step's ONE collective: the all-reduce of the flat fp32 gradient bucket the reference's two
Adam groups cover (methods/DKT.py:114-115: backbone + bn_out of the config, and the 2C GP hyper-parameters).
The hyper-parameter gradients in the bucket are the real ones of the step; the backbone part is a resident
buffer of the config's size (the backbone itself runs in MIOpen and is not part of the hot path).
Workload = BASELINE.json configs[1]-class headline "5-way 5-shot Conv4" shape the metric is quoted on
(N=105, D=1600, C=5); inputs (normalised features Z) are resident in HBM before the timed
region.  Episodes shard across ranks with no data-path collective (weak scaling).

  python bench.py [--gpus N --steps K --warmup W --episodes B --config cfg2]
      N > 1 without a launcher: bench.py spawns the N ranks itself (one process per GPU, RCCL).
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Timing: W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + torch.cuda.synchronize()
on both sides, MAX over ranks; blocks repeat until >= 1 s has been timed (at least 3) and the MEDIAN block is
reported (`timing` holds min / max / count).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32-input MFMA = fp32 vector peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16 / bf16 MFMA

# name -> (n_way, n_support, n_query, D, description, floats in the backbone + bn_out gradient bucket)
CONFIGS = {
    # QMUL head-pose regression (the reference's CPU-runnable case, train_regression.py): ONE GP per task, 19 frames, Conv3 features,
    # RBF kernel, learned noise (DKT_regression.py:45-64, qmul_loader.py); bucket = Conv3 (24 408) + 4 GP hyper-parameters
    "cfg0": (1, 5, 14, 2916, "QMUL regression, Conv3 features, RBF kernel: one task = one (19, 2916) GP", 24408),
    "cfg1": (5, 5, 16, 64, "Omniglot 5-way 5-shot, Conv4S features", 112064),
    # Omniglot 20-way (train.py:85-93 forces Conv4S: D = 64; --train_n_way 20: N = 420): not a BASELINE config; the shape where the feature-space path
    # (ops.lowrank_applies: twenty 64 x 64 models instead of twenty 420 x 420 ones) matters most
    "cfg1_20way": (20, 5, 16, 64, "Omniglot 20-way 5-shot, Conv4S features (N = 420, D = 64: feature-space path)", 112064),
    "cfg2": (5, 5, 16, 1600, "CUB 5-way 5-shot, Conv4 features (headline)", 116288),
    "cfg3": (5, 1, 16, 512, "miniImagenet 5-way 1-shot, ResNet10 features", 4906816),
    # 20-way: N = 420 is what train_loop builds (20 x (5 + 16)); BASELINE.json quotes a 320 x 320 Gram
    "cfg4": (20, 5, 16, 512, "miniImagenet 20-way 5-shot, ResNet18 features (N = 420: band-reduction MLL path from 192 episodes per call)", 11177536),
    "cfg4_n320": (20, 1, 15, 512, "20-way, N = 320 (the Gram size BASELINE.json quotes)", 11177536),
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


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        return os.cpu_count() or 1


def _cgroup_cpu_quota():
    """CPUs' worth of time the container's cgroup grants (cpu.max = "<quota> <period>" on cgroup v2, cfs_quota_us / cfs_period_us on v1);
    None = unlimited.  psutil / os.cpu_count() see the HOST's cores and ignore this."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _usable_cores():
    """(cpu ids this process may run on -- one per PHYSICAL core where the SMT topology is readable --, facts dict).  The all-cores baseline
    starts one pinned single-thread process per entry, capped by the cgroup quota."""
    try:
        aff = sorted(os.sched_getaffinity(0))
    except AttributeError:
        aff = list(range(os.cpu_count() or 1))
    seen, cpus = set(), []
    for cpu in aff:                                   # one logical CPU per physical core (SMT siblings share the FP pipes)
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu).read().strip()
        except OSError:
            sib = str(cpu)
        if sib not in seen:
            seen.add(sib)
            cpus.append(cpu)
    quota = _cgroup_cpu_quota()
    facts = {"host_cpus": os.cpu_count(), "affinity_cpus": len(aff), "physical_cores_in_affinity": len(cpus), "physical_cores_psutil": _physical_cores(),
             "cgroup_cpu_quota": quota}
    if quota is not None:
        cpus = cpus[:max(1, int(quota))]
    facts["cores_effective"] = len(cpus)
    return cpus, facts


def _cpu_worker(job):
    """One host process = one core: single-threaded torch, B = 1 sequential training episodes (the way the reference runs them) of the
    same synthetic workload, counted inside wall-clock windows that every worker shares (so the sum over the workers is a rate all
    cores sustained AT THE SAME TIME)."""
    wid, cpu, n, d, n_way, raw_s, mean, t_open_v, ready_q, windows, window_s = job
    try:
        os.sched_setaffinity(0, {cpu})               # one process per core, pinned (the parent exported OMP / MKL_NUM_THREADS = 1 before the spawn)
    except (AttributeError, OSError):
        pass
    import torch as _t
    _t.set_num_threads(1)
    from oracle import dkt_oracle_torch as T
    g = _t.Generator().manual_seed(5000 + wid)
    z = _t.randn(8, n, d, generator=g)
    z = (z - z.mean(1, keepdim=True)) / _t.sqrt(z.var(1, unbiased=False, keepdim=True) + 1e-5)
    raw_s, mean = _t.tensor(raw_s), _t.tensor(mean)

    def episode(i):
        zi = z[i % z.shape[0]].clone().requires_grad_(True)
        T.cpu_baseline_train_episode(zi, n_way, raw_s.clone().requires_grad_(True), mean.clone().requires_grad_(True))
    for i in range(3):
        episode(i)
    ready_q.put(wid)                             # imported, warmed up: the parent opens the windows once every worker said so
    i = 3
    while t_open_v.value == 0.0 or time.time() < t_open_v.value:      # keep running (warm) until the first window opens
        episode(i)
        i += 1
        if t_open_v.value < 0.0:
            return []
    t_open = t_open_v.value
    counts = []
    for wdw in range(windows):
        t_end = t_open + (wdw + 1) * window_s
        c = 0
        while time.time() < t_end:
            episode(i)
            i += 1
            c += 1
        counts.append(c)
    return counts


def _cpu_worker_entry(job, q):
    try:
        q.put(_cpu_worker(job))
    except Exception as e:  # noqa: BLE001
        q.put("worker %d failed: %r" % (job[0], e))


class _QuietChildren:
    """Children started inside this block inherit /dev/null as stdout / stderr, single-thread OpenMP / MKL pools and NO visible GPU: the
    baseline processes are host-only, and their libdrm / HIP start-up chatter must never reach the bench's stdout (the JSON line is the
    last -- and only -- line there) or fill the driver's stderr tail."""
    ENV = {"OMP_NUM_THREADS": "1", "MKL_NUM_THREADS": "1", "OPENBLAS_NUM_THREADS": "1", "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": "",
           "CUDA_VISIBLE_DEVICES": ""}

    def __enter__(self):
        sys.stdout.flush()
        sys.stderr.flush()
        self.saved_env = {k: os.environ.get(k) for k in self.ENV}
        os.environ.update(self.ENV)
        self.fds = (os.dup(1), os.dup(2))
        nul = os.open(os.devnull, os.O_WRONLY)
        os.dup2(nul, 1)
        os.dup2(nul, 2)
        os.close(nul)
        return self

    def __exit__(self, *exc):
        os.dup2(self.fds[0], 1)
        os.dup2(self.fds[1], 2)
        os.close(self.fds[0])
        os.close(self.fds[1])
        for k, v in self.saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        return False


def cpu_baseline(z_cpu, n_way, raw_s, mean, warm=20, timed=200, repeats=5, windows=3, window_s=3.0):
    """BASELINE.md section 4 protocol: the oracle's fp32 GPyTorch-structured port (per-class loop, dense Cholesky, autograd
    backward), B = 1 sequential episodes as the reference runs them, on this box's host cores:
      * `1`: one process, one thread: `warm` untimed + `timed` timed episodes, median of up to `repeats` (<= 10 s);
      * `all_cores`: P = `cores_effective` independent single-thread PROCESSES (one per physical core of this process's affinity mask,
        capped by the cgroup CPU quota; each pinned to its core, OMP / MKL_NUM_THREADS = 1), each running its own sequential episode
        stream; the episodes all of them complete inside `windows` shared wall-clock windows of `window_s` seconds, summed (median
        window).  This is the fair "all host cores" denominator: one B = 1 stream cannot use 128 threads (round 3 timed exactly that and
        got 22 x LESS than one thread)."""
    from oracle import dkt_oracle_torch as T

    def episode(i):
        zi = z_cpu[i % z_cpu.shape[0]].clone().requires_grad_(True)
        T.cpu_baseline_train_episode(zi, n_way, raw_s.clone().requires_grad_(True), mean.clone().requires_grad_(True))

    res = {}
    cpus, facts = _usable_cores()
    res["cores"] = facts
    nproc = len(cpus)
    torch.set_num_threads(1)
    for i in range(warm):
        episode(i)
    rates, t_start = [], time.perf_counter()
    for _ in range(repeats):
        t0 = time.perf_counter()
        for i in range(timed):
            episode(i)
        rates.append(timed / (time.perf_counter() - t0))
        if time.perf_counter() - t_start > 10.0:          # bounded: the default run must stay within minutes
            break
    res["1"] = dict(episodes_per_s=round(statistics.median(rates), 2), repeats=len(rates), processes=1, threads_per_process=1)
    torch.set_num_threads(min(nproc, 8))
    if nproc > 1:
        try:
            import multiprocessing as mp
            ctx = mp.get_context("spawn")
            n, d = int(z_cpu.shape[1]), int(z_cpu.shape[2])
            t_open_v = ctx.Value("d", 0.0)
            q, ready_q = ctx.Queue(), ctx.Queue()
            jobs = [(w, cpus[w], n, d, n_way, raw_s.tolist(), mean.tolist(), t_open_v, ready_q, windows, window_s) for w in range(nproc)]
            procs = [ctx.Process(target=_cpu_worker_entry, args=(j, q), daemon=True) for j in jobs]
            with _QuietChildren():
                for pr in procs:
                    pr.start()
            counts = []
            try:
                t_lim = time.time() + 90.0                   # every worker imports torch first: the windows open when ALL of them are warm
                nready = 0
                while nready < nproc and time.time() < t_lim:
                    try:
                        ready_q.get(timeout=1.0)
                        nready += 1
                    except Exception:  # noqa: BLE001 -- queue.Empty
                        pass
                if nready < nproc:
                    t_open_v.value = -1.0
                    raise RuntimeError("only %d of %d baseline processes came up within 90 s" % (nready, nproc))
                t_open_v.value = time.time() + 1.0
                deadline = t_open_v.value + windows * window_s + 30.0
                while len(counts) < nproc:
                    counts.append(q.get(timeout=max(1.0, deadline - time.time())))
            finally:                                         # (never a hang: whatever has not answered by the deadline is killed)
                for pr in procs:
                    pr.join(timeout=0.5)
                    if pr.is_alive():
                        pr.kill()
            if any(isinstance(c, str) for c in counts):
                raise RuntimeError([c for c in counts if isinstance(c, str)][0])
            per_window = [sum(c[wdw] for c in counts) / window_s for wdw in range(windows)]
            allc = statistics.median(per_window)
            res["all_cores"] = dict(episodes_per_s=round(allc, 2), repeats=windows, processes=nproc, threads_per_process=1,
                                    window_s=window_s, per_window=[round(v, 1) for v in per_window],
                                    slowest_process_eps=round(min(statistics.median(c) for c in counts) / window_s, 2),
                                    fastest_process_eps=round(max(statistics.median(c) for c in counts) / window_s, 2),
                                    parallel_efficiency=round(allc / (nproc * res["1"]["episodes_per_s"]), 3))
        except Exception as e:  # noqa: BLE001 -- the baseline is a report, never a reason to lose the bench line
            res["all_cores"] = dict(error=repr(e)[:200])
    return res


def gpytorch_baseline(z_cpu, n_way, raw_s, mean, episodes=50):
    """The genuine reference objects (methods/DKT.py:58-71, 337-378) on CPU, if GPyTorch happens to be importable on this box
    (it is not in the build image): IndependentModelList + SumMarginalLogLikelihood of ExactGPs with ScaleKernel(LinearKernel).
    Returns None when the import fails."""
    try:
        import gpytorch
    except Exception:
        return None
    import torch.nn.functional as F

    class Layer(gpytorch.models.ExactGP):
        def __init__(self, x, y, lik):
            super().__init__(x, y, lik)
            self.mean_module = gpytorch.means.ConstantMean()
            self.covar_module = gpytorch.kernels.ScaleKernel(gpytorch.kernels.LinearKernel())
            self.covar_module.base_kernel.variance = 1.0
            self.covar_module.base_kernel.raw_variance.requires_grad = False

        def forward(self, x):
            return gpytorch.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    n = z_cpu.shape[1]
    models, liks = [], []
    for c in range(n_way):
        lik = gpytorch.likelihoods.GaussianLikelihood()
        lik.noise = 0.1
        lik.raw_noise.requires_grad = False
        m = Layer(torch.ones(n, z_cpu.shape[2]), torch.ones(n), lik)
        m.covar_module.raw_outputscale.data.fill_(float(raw_s[c]))
        m.mean_module.constant.data.fill_(float(mean[c]))
        models.append(m)
        liks.append(lik)
    model = gpytorch.models.IndependentModelList(*models)
    mll = gpytorch.mlls.SumMarginalLogLikelihood(gpytorch.likelihoods.LikelihoodList(*liks), model)
    model.train()
    per = n // n_way
    cls = torch.arange(n_way).repeat_interleave(per)
    ys = [torch.where(cls == c, 1.0, -1.0) for c in range(n_way)]

    def episode(i):
        z = F.normalize(z_cpu[i % z_cpu.shape[0]].clone().requires_grad_(True), p=2, dim=1)
        for m, y in zip(model.models, ys):
            m.set_train_data(inputs=z, targets=y, strict=False)
        loss = -mll(model(*model.train_inputs), model.train_targets)
        loss.backward()
        return float(loss)

    torch.set_num_threads(1)
    loss0 = episode(0)
    for i in range(5):
        episode(i)
    t0 = time.perf_counter()
    for i in range(episodes):
        episode(i)
    return dict(value=round(episodes / (time.perf_counter() - t0), 2), unit="episodes/s", cores=1, version=gpytorch.__version__, loss_episode0=loss0)


def _workload(cfg, b, dev, rank, unit_rows):
    """Resident inputs + the step closure of one BASELINE config.  Returns (step, state): step() runs one pass of the hot path
    (forward + backward) and returns (loss, logp, info); state holds the tensors the checks and the collective need."""
    from dkt_amd import ops
    c, s, q, d, desc, n_backbone = CONFIGS[cfg]
    n = c * (s + q)
    raw_s, mean = perturbed_hypers(c, 99, dev)
    raw_s.requires_grad_(True)
    mean.requires_grad_(True)
    if cfg == "cfg0":
        # QMUL regression head (DKT_regression.py:45-64): one GP per task on Conv3 features [19, 2916], RBF kernel, learned noise;
        # features ~ ReLU-like non-negative activations, targets = head-pose angles scaled to [-1, 1] per task ([B, 1, N])
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        z = (torch.randn(b, n, d, generator=g, device=dev).abs() * 0.35).contiguous().requires_grad_(True)
        y = (torch.rand(b, 1, n, generator=g, device=dev) * 2.0 - 1.0).contiguous()
        raw_ls = torch.tensor([11.0], device=dev, requires_grad=True)          # lengthscale = softplus(raw) ~ 11: off-diagonal kernel values ~ 0.3 at this
                                                                               # feature scale (round 3 ran 2.58: E = I to 6e-7, a degenerate task)
        raw_nz = torch.tensor([0.0], device=dev, requires_grad=True)           # GaussianLikelihood default: noise = softplus(0) + 1e-4
        cw = torch.full((1,), -1.0 / n, device=dev)
        params = [raw_s, mean, raw_ls, raw_nz]

        def step():
            z.grad = None
            for p_ in params:
                p_.grad = None
            sv = torch.nn.functional.softplus(raw_s)
            ls = torch.nn.functional.softplus(raw_ls)
            nz = torch.nn.functional.softplus(raw_nz) + 1e-4
            e = ops.base_matrix(z, "rbf", ls)
            obj, logp, alpha, info, jit = ops.mll_objective(e, y, sv, mean, nz, cw)
            loss = obj.mean()
            loss.backward()
            return loss, logp, info
        return step, dict(z=z, y=y, raw_s=raw_s, mean=mean, params=params, c=c, s=s, q=q, n=n, d=d, desc=desc, n_backbone=n_backbone, kernel="rbf")
    z = synthetic_batch(b, n, d, 1234 + rank, dev).requires_grad_(True)
    noise = torch.full((c,), 0.1, device=dev)
    cls = torch.arange(c, device=dev).repeat_interleave(s + q)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    cw = torch.full((c,), -1.0 / (c * n), device=dev)

    def step():
        z.grad = None
        raw_s.grad = None
        mean.grad = None
        sv = torch.nn.functional.softplus(raw_s)
        # the features are bn_out'ed + L2-normalised rows (the cossim / bncossim contract): unit_rows lets the Gram kernels
        # take the scaled 2-way f16 split; DKT_BENCH_UNIT=0 runs the range-agnostic 3-way bf16 split instead
        obj, logp, alpha, info, jit, e = ops.episode_loss_linear(z, y, sv, mean, noise, cw, unit_rows=unit_rows)
        loss = obj.mean()
        loss.backward()
        return loss, logp, info
    return step, dict(z=z, y=y, raw_s=raw_s, mean=mean, noise=noise, params=[raw_s, mean], c=c, s=s, q=q, n=n, d=d, desc=desc,
                      n_backbone=n_backbone, kernel="bncossim")


def _aux_paths(dev, cfg="cfg2", b=2048, b_rbf=None, steps=20, only=None):
    """The two other entries into the same hot path, driver-timed next to the graded one, at the headline shape (cfg2: N = 105, D = 1600, C = 5) and at the
    20-way shape (cfg4: N = 420, D = 512, C = 20):
    `from_trunk_features` = what DKT.train_loop runs for bncossim -- N <= 128: bn_out in train mode + F.normalize folded into the Gram kernels
    (dkt_gram_bn_train_f32 -> dkt_mll_f32 -> dkt_gram_bn_bwd_f32); N > 128: dkt_bn_stats_f32 -> dkt_affine_normalize_f32 -> dkt_gram_f32 -> dkt_mll_f32 ->
    dkt_gram_bwd_f32 -> dkt_normalize_bn_bwd_f32 --,
    `rbf_per_class_lengthscales` = the non-linear kernels (dkt_gram_f32 SQDIST -> dkt_class_kernel_f32 -> dkt_mll_f32 with DKT_MLL_E_PER_CLASS ->
    dkt_class_kernel_bwd_f32 -> dkt_gram_bwd_f32)."""
    from dkt_amd import ops
    c, s, q, d = CONFIGS[cfg][:4]
    n = c * (s + q)
    b_rbf = b if b_rbf is None else b_rbf
    g = torch.Generator(device=dev).manual_seed(4321)
    x = (torch.randn(b, n, d, generator=g, device=dev).abs() + 1.0).requires_grad_(True)        # ReLU-like trunk output, common offset
    gamma = torch.ones(d, device=dev, requires_grad=True)
    beta = torch.zeros(d, device=dev, requires_grad=True)
    raw_s, mean = perturbed_hypers(c, 99, dev)
    raw_s.requires_grad_(True)
    mean.requires_grad_(True)
    ls = (torch.linspace(25.0, 40.0, c, device=dev) * (d / 1600.0) ** 0.5).requires_grad_(True)   # ~ the distance scale of these features
    noise = torch.full((c,), 0.1, device=dev)
    cls = torch.arange(c, device=dev).repeat_interleave(s + q)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    x_rbf = x if b_rbf == b else x[:b_rbf].detach().clone().requires_grad_(True)               # (a leaf of its own: a slice inside the step would time autograd's zero-padding)
    leaves = (x, x_rbf, gamma, beta, raw_s, mean, ls)

    def trunk():
        for t in leaves:
            t.grad = None
        outs = ops.episode_loss_bn(x, gamma, beta, y, torch.nn.functional.softplus(raw_s), mean, noise, cw)
        outs[0].mean().backward()
        return outs[3]

    def rbf():
        for t in leaves:
            t.grad = None
        outs = ops.episode_loss_class_kernel(x_rbf, y, torch.nn.functional.softplus(raw_s), mean, noise, cw, "rbf", ls)
        outs[0].mean().backward()
        return outs[3]

    res = {}
    for name, fn, nb in (("from_trunk_features", trunk, b), ("rbf_per_class_lengthscales", rbf, b_rbf)):
        if only is not None and name != only:
            continue
        for _ in range(max(2, steps // 2)):                   # (the first steps of a fresh shape run slower: allocator growth, clock ramp -- tools/glue_probe.py)
            info = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            info = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        # the per-kernel HIP events in a pass of their own (they are required inside the timed region of the graded step only; here each pair costs the
        # 1.3-ms step of 2048 episodes ~ 0.03 ms: tools/glue_probe.py)
        ops.kernel_timing(True)
        for _ in range(steps):
            info = fn()
        torch.cuda.synchronize()
        kt = {k: round(v[1], 4) for k, v in ops.kernel_timing_results().items()}
        ops.kernel_timing(False)
        # algorithmic bytes per episode of the calls of these paths (include/dkt_abi.h contracts: every operand once) -> HBM roofline per kernel
        nn, nd = 4 * n * n, 4 * n * d
        alg_b = {"dkt_gram_bn_train_f32": nd + nn + 4 * n + 20 * d, "dkt_gram_bn_bwd_f32": 2 * nn + 2 * nd + 4 * n + 24 * d, "dkt_bn_stats_f32": nd + 20 * d,
                 "dkt_affine_normalize_f32": 2 * nd + 4 * n + 8 * d, "dkt_normalize_bn_bwd_f32": 4 * nd + 4 * n + 24 * d, "dkt_gram_f32": nd + nn,
                 "dkt_gram_bwd_f32": nn + 2 * nd, "dkt_class_kernel_f32": nn + c * nn, "dkt_class_kernel_bwd_f32": c * nn + 2 * nn,
                 "dkt_lowrank_gram_f32": nd + 4 * 64 * (64 + c), "dkt_lowrank_finish_f32": nd + 4 * c * (64 + 2 * n), "dkt_lowrank_bwd_f32": 2 * nd + 4 * 64 * 64 + 4 * c * (n + 64)}
        roofs = {k: {"bound": "hbm", "achieved": round(alg_b[k] * nb / v / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg_b[k] * nb / v / 1e6 / HBM_PEAK_GBS, 4)} for k, v in kt.items() if k in alg_b and v > 0}
        # PMC traffic (FETCH x 2 + WRITE) of the fused front-end kernels from the committed profile of this path, scaled to this batch
        tjf = _traffic_table().get(cfg + "_from_trunk") if name == "from_trunk_features" else None
        if tjf:
            for k, r in roofs.items():
                if k in tjf["kernels"]:
                    r["traffic"] = round(tjf["kernels"][k]["hbm_bytes"] * nb / tjf["episodes_per_launch"])
                    r["traffic_over_algorithmic"] = round(r["traffic"] / (alg_b[k] * nb), 3)
                    r["traffic_source"] = tjf["source"]
        res[name] = {"value": round(nb / dt, 1), "unit": "episodes/s", "episodes_per_step": nb, "ms_per_step": round(1e3 * dt, 4), "roofline": roofs,
                     "valid": bool(int(info.abs().max().item()) == 0 and (x_rbf if fn is rbf else x).grad is not None
                                   and bool(torch.isfinite((x_rbf if fn is rbf else x).grad).all().item())),
                     "kernels_ms": kt}
    return res


GLUE_KERNELS = ("dkt_objective_f32", "dkt_hyper_grads_f32", "dkt_bn_param_grads_f32")


def _algorithmic(cfg, n, d, c, unit_rows, lowrank=False):
    """Algorithmic bytes / flops per EPISODE, per ABI kernel of the step; `exec_f16`: the f16 MFMA flops the split Gram
    kernels actually execute (lower 16 x 16 tiles x 3 products of the scaled 2-way split; 6 for the bf16 split).
    `lowrank` (D <= 64 < N, ops.lowrank_applies): the step is dkt_lowrank_gram_f32 -> dkt_mll_f32 on the 64 x 64 models -> dkt_lowrank_finish_f32 ->
    dkt_lowrank_bwd_f32; its bytes are what those calls have to move (Z three times, dZ once, the 64 x 64 matrices A and W' once each way)."""
    glue = {   # the two [B, C]-sized reductions around the marginal likelihood (dkt_objective.hip): launches, no bytes to speak of
        "dkt_objective_f32": dict(bytes=4 * (c + 1), flops=2 * c, exec_f16=None),
        "dkt_hyper_grads_f32": dict(bytes=4 * (2 * c + 1), flops=4 * c, exec_f16=None),
    }
    if lowrank:
        dp = 64
        return {
            **glue,
            "dkt_lowrank_gram_f32": dict(bytes=4 * (n * d + dp * dp + c * dp), flops=2 * n * d * (d + c), exec_f16=None),
            "dkt_mll_f32": dict(bytes=4 * (2 * dp * dp + 3 * c * dp), flops=c * (dp ** 3 // 3 + 2 * dp * dp + dp ** 3), exec_f16=None),
            "dkt_lowrank_finish_f32": dict(bytes=4 * (n * d + c * dp + 2 * c * n), flops=2 * n * d * c, exec_f16=None),
            "dkt_lowrank_bwd_f32": dict(bytes=4 * (2 * n * d + dp * dp + c * n + c * dp), flops=2 * n * d * (d + c), exec_f16=None),
        }
    nt16 = (n + 15) // 16
    nprod = 3 if unit_rows else 6
    alg = {
        **glue,
        "dkt_gram_f32": dict(bytes=4 * (n * d + n * n), flops=2 * n * n * d,
                             exec_f16=(nt16 * (nt16 + 1) // 2) * nprod * 2 * 256 * d if (n <= 128 and cfg != "cfg0") else None),
        "dkt_mll_f32": dict(bytes=4 * (2 * n * n + 3 * c * n), flops=c * (n ** 3 // 3 + 2 * n * n + n ** 3), exec_f16=None),
        "dkt_gram_bwd_f32": dict(bytes=4 * (n * n + 2 * n * d), flops=2 * n * n * d,
                                 exec_f16=nt16 * nt16 * nprod * 2 * 256 * d if (n <= 128 and cfg != "cfg0") else None),
        "dkt_rbf_bwd_f32": dict(bytes=4 * 3 * n * n, flops=6 * n * n, exec_f16=None),      # reads W and E, writes W' (+ the d lengthscale reduction)
    }
    return alg


def _measure(cfg, b, args, dev, rank, world, bucket_fn, unit_rows, min_blocks, min_total, steps, ktime=True):
    """Warm-up, then blocks of EXACTLY `steps` steps bracketed by barrier + synchronize on both sides (MAX over ranks), at least
    `min_blocks` blocks and `min_total` seconds; per-kernel HIP-event times from the same blocks.  Returns the measurement dict."""
    from dkt_amd import ops
    step0, st = _workload(cfg, b, dev, rank, unit_rows)
    bucket = bucket_fn(st)

    def step():
        out = step0()
        if bucket is not None:
            bucket.allreduce_mean()      # the path's only exchange: the shared-parameter gradient bucket (no-op at world 1)
        return out

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    blocks, total, ktimes_all = [], 0.0, {}
    while len(blocks) < min_blocks or (total < min_total and len(blocks) < 50):
        ops.kernel_timing(ktime)                 # (two HIP events per ABI call: ~ 10 us of host time each -- off for the launch-bound small-batch rows)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, logp, info = step()
        sync()
        dt = time.perf_counter() - t0
        for name, (cnt, ms) in ops.kernel_timing_results().items():
            ktimes_all.setdefault(name, []).append((cnt, ms))
        ops.kernel_timing(False)
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        blocks.append(dt)
        total += dt
    dt = statistics.median(blocks)
    # per kernel: launches over all blocks, and the MEDIAN over the blocks of the block's average launch duration (a one-off stall in
    # one block -- an allocation, a clock ramp after the previous config -- does not leak into the reported kernel time)
    ktimes = {name: (sum(cn for cn, _ in v), statistics.median(ms for _, ms in v)) for name, v in ktimes_all.items()}
    ok = int(info.abs().max().item()) == 0 and bool(torch.isfinite(loss))
    # same inputs -> bitwise the same outputs (no atomics, fixed reduction orders): a hand-off race in a kernel would
    # show up here as a handful of differing episodes
    _, logp_a, _ = step()
    grad_a = st["z"].grad.clone()
    _, logp_b, _ = step()
    deterministic = bool(torch.equal(logp_a, logp_b)) and bool(torch.equal(grad_a, st["z"].grad))
    return dict(st=st, bucket=bucket, step=step, dt=dt, blocks=blocks, total=total, ktimes=ktimes, valid=ok and deterministic,
                deterministic=deterministic, logp=logp, steps=steps, b=b)


def _traffic_table():
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (bench.py cannot profile itself), per config
    (profiles/pmc_traffic.json, written by tools/make_traffic_json.py from tools/prof_all.sh's summaries)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return {}
    if "configs" in tj:
        return tj["configs"]
    return {tj.get("config", "cfg2"): tj}


def _kernel_report(cfg, m, unit_rows, traffic):
    from dkt_amd import ops
    st, b = m["st"], m["b"]
    n, d, c = st["n"], st["d"], st["c"]
    lowrank = st["kernel"] == "bncossim" and ops.lowrank_applies(n, d, c, b)
    alg = _algorithmic(cfg, n, d, c, unit_rows, lowrank)
    if lowrank:
        n = ops.LOWRANK_DP                    # the size the marginal-likelihood kernel runs at (for its executed-flop count below)
    kernels = {}
    for name, (cnt, ms) in m["ktimes"].items():
        a = alg[name]
        kernels[name] = dict(launches=cnt, ms=round(ms, 4), gbs=round(a["bytes"] * b / ms / 1e6, 1), tflops=round(a["flops"] * b / ms / 1e9, 2))
    tj = traffic.get(cfg)

    def roof(name):
        """Both roofs for one kernel.  The Gram kernels stream Z once: HBM binds them (their contraction runs on the f16 pipe,
        `executed_f16_mfma`).  The marginal-likelihood kernel has no HBM pressure: its algorithmic fp32 flops against the fp32 matrix
        peak (N <= 127: the factorisation and the K^-1 product themselves run as 2-way f16 splits on the f16 pipe since round 3)."""
        k = kernels[name]
        tr, src, tr_note = None, None, None
        if tj and name in tj.get("kernels", {}):
            tr = round(tj["kernels"][name]["hbm_bytes"] * b / tj["episodes_per_launch"])
            src = tj["source"]
            if tr > k["ms"] * 1e-3 * HBM_PEAK_GBS * 1e9:           # more bytes than the HBM peak could move in the measured time: not evidence
                tr_note = "profile figure %d B rejected: it exceeds %.1f ms x %.0f GB/s" % (tr, k["ms"], HBM_PEAK_GBS)
                tr = None
        hbm = dict(bound="hbm", achieved=k["gbs"], peak=HBM_PEAK_GBS, unit="GB/s", frac=round(k["gbs"] / HBM_PEAK_GBS, 4))
        mat = dict(bound="mfma", achieved=k["tflops"], peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(k["tflops"] / MFMA_F32_PEAK_TFLOPS, 4),
                   note="algorithmic fp32 flops / fp32 MFMA peak")
        first, other = (mat, hbm) if name == "dkt_mll_f32" else (hbm, None)
        if name == "dkt_mll_f32" and n + 1 <= 128:
            # N <= 127: every tile product of the factorisation / inverse / K^-1 runs as 3 v_mfma_f32_16x16x16_f16 (8192 flop each) on the
            # f16 pipe: executed products per class matrix = 2 NT(NT^2-1)/6 + NT(NT-1) + NT (phases 1-2) + NT(NT+1)(NT+2)/6 (phase 3)
            # (NT = 7: 245; the SQ_INSTS_MFMA counter of profiles/r03/cfg2_summary.txt gives 246).  That pipe's dense peak is the roof.
            nt = (n + 1 + 15) // 16
            prods = 2 * nt * (nt * nt - 1) // 6 + nt * (nt - 1) + nt + nt * (nt + 1) * (nt + 2) // 6
            ex = prods * 3 * 8192.0 * c * b / k["ms"] / 1e9
            first = dict(bound="mfma", achieved=round(ex, 1), peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ex / MFMA_F16_PEAK_TFLOPS, 4),
                         note="EXECUTED v_mfma_f32_16x16x16_f16 flops (%d tile products x 3 plane products per class matrix) / dense f16 peak; the kernel is "
                              "bound by VALU issue + the dependency chain of the diagonal-tile sweeps, not by this pipe (DESIGN.md 4.2)" % prods)
            other = dict(mat, note="algorithmic fp32-equivalent flops / fp32 MFMA peak (the pipe rounds 1-2 used; kept for comparison across rounds)")
        band = name == "dkt_mll_f32" and n >= 128 and (n + 15) // 16 <= 27 and 12 <= c <= 32 and b >= 192 and st["kernel"] in ("bncossim", "cossim", "linear")
        if band:
            # band reduction (round 6, csrc/dkt_mll_band.hip): ONE orthogonal reduction of E per episode (fp32 MFMA: Householder panels + fused two-sided passes over the
            # lower block triangle), C block-LDL^T chains + column chains (fp32 MFMA), one similarity transform back (2-way f16 splits).  Executed MFMA work per episode:
            #   forward   passes it = 0 .. NT-2 over the stored tiles (i >= j >= it + 1): 16 fp32 instructions per off-diagonal tile (8 update + 4 + 4 for X'), 12 per diagonal
            #   back      rows >= the panel's first row, every column: 12 f16 instructions (6 + 3 + 3) per off-diagonal tile
            #   chain     C nt (nt + 1) / 2 tile products (4 fp32 instructions), class kernel ~ 14 products per block and class
            # against the fp32 matrix peak resp. the dense f16 peak.  The kernels are bound by neither pipe nor by HBM: by the per-episode chain of 2 x (NT - 2) panels with
            # two workgroups per CU (phase clocks: profiles/r06/band_phase_clocks.log) -- `fabric_gbs` / the executed fractions say how far from each roof.
            nt = (n + 15) // 16
            fwd_tiles = sum((nt - jl) * (nt - jl + 1) // 2 for jl in range(1, nt))
            back_tiles = sum(sum(i + 1 for i in range(r, nt)) for r in range(1, nt - 1))
            ex32 = (fwd_tiles * 16 + c * (nt * (nt + 1) // 2 * 4 + nt * 14 * 4)) * 2048.0 * b / k["ms"] / 1e9
            ex16 = back_tiles * 12 * 8192.0 * b / k["ms"] / 1e9
            real = (fwd_tiles * 4 + back_tiles * 4 + c * (nt * (nt + 1) // 2 + nt * 14)) * 8192.0          # tile products x 2 x 16^3
            first = dict(bound="mfma", achieved=round(ex32, 1), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ex32 / MFMA_F32_PEAK_TFLOPS, 4),
                         note="band reduction: EXECUTED v_mfma_f32_16x16x4_f32 flops (forward passes, class + column chains) / fp32 matrix peak; the back transform's f16 "
                              "split products are in executed_f16; the path does %.1f x fewer flops than the C factorisations + inverses `algorithmic_flops_per_launch` "
                              "counts (the reference algorithm)" % (alg[name]["flops"] / real))
            first["executed_f16_tflops"] = round(ex16, 1)
            first["executed_f16_frac"] = round(ex16 / MFMA_F16_PEAK_TFLOPS, 4)
            first["path"] = "band"
            other = dict(mat, note="algorithmic fp32 flops of the REFERENCE algorithm (C factorisations + inverses) per second / fp32 MFMA peak -- an equivalent rate, "
                                    "not executed work (kept for comparison with the tile-array rounds)")
            if tr:
                first["fabric_gbs"] = round(tr / k["ms"] / 1e6, 1)
                first["fabric_frac"] = round(tr / k["ms"] / 1e6 / HBM_PEAK_GBS, 4)
        if name == "dkt_mll_f32" and n + 1 > 128 and not band:
            # tile-array path (127 < N <= 447): since round 4 the left-looking K loops of the factorisation / inverse and the whole K^-1 product run as
            # f16-split tile products (3 v_mfma_f32_16x16x16_f16 each), the in-block panels / updates and the diagonal sweeps in fp32.  The kernels are
            # bound by their tile streams (traffic-only builds: profiles/r04/v0_tiled_traffic_ceiling.txt), not by a matrix pipe: `fabric_gbs` = PMC
            # bytes at the L2 <-> fabric boundary / time.
            first = dict(bound="mfma", achieved=k["tflops"], peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s", frac=round(k["tflops"] / MFMA_F16_PEAK_TFLOPS, 4),
                         note="algorithmic fp32-equivalent flops / dense f16 MFMA peak (the pipe the K loops and the K^-1 product run on); the kernels are bound by "
                              "their tile streams, see traffic / fabric_gbs")
            other = dict(mat, note="algorithmic fp32-equivalent flops / fp32 MFMA peak (the pipe rounds 1-3 used for the factorisation / inverse; kept for comparison)")
            # executed f16 MFMA flops of the three K loops: tile products per class matrix (factor sum_{i<=j} i, invert sum_{i<j} (j - i), W sum_{i<=j} (NT - j)) x 3 plane
            # products x 8192 flop, against the dense f16 peak -- next to the fabric bytes per second against the HBM peak: the kernels are bound by the latter
            nt = (n + 1 + 15) // 16
            prods = sum(i for j in range(nt) for i in range(j + 1)) + sum(j - i for j in range(nt) for i in range(j)) + sum(nt - j for j in range(nt) for i in range(j + 1))
            ex = prods * 3 * 8192.0 * c * b / k["ms"] / 1e9
            first["executed_tflops"] = round(ex, 1)
            first["executed_frac"] = round(ex / MFMA_F16_PEAK_TFLOPS, 4)
            if tr:
                first["fabric_gbs"] = round(tr / k["ms"] / 1e6, 1)
                first["fabric_frac"] = round(tr / k["ms"] / 1e6 / HBM_PEAK_GBS, 4)
        r = dict(kernel=name, **first, traffic=tr, traffic_unit="bytes/launch (PMC: FETCH_SIZE + WRITE_SIZE at L2<->fabric)",
                 traffic_source=src, algorithmic_bytes_per_launch=alg[name]["bytes"] * b,
                 algorithmic_flops_per_launch=alg[name]["flops"] * b, avg_launch_ms=k["ms"], episodes_per_launch=b)
        if other:
            r["other_roof"] = other
        if tr_note:
            r["traffic_note"] = tr_note
        if alg[name]["exec_f16"]:
            ex = alg[name]["exec_f16"] * b / k["ms"] / 1e9
            r["executed_f16_mfma"] = dict(achieved=round(ex, 1), peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ex / MFMA_F16_PEAK_TFLOPS, 4),
                                          note="executed v_mfma_f32_16x16x32_f16 flops (split products, computed tiles only) / dense f16 peak")
        return r
    return kernels, {name: roof(name) for name in kernels}


def _default_batch(cfg):
    c, s, q = CONFIGS[cfg][:3]
    return 8192 if c * (s + q) <= 128 else (2048 if cfg == "cfg1_20way" else 1024)


def _claim_stdout():
    """The graded contract: ONE JSON line, the last (here: the only) thing on stdout.  Libraries write to fd 1 behind Python's back (RCCL's
    version banner through C stdio, flushed at exit -- round 4 lost its bench record to exactly that; libdrm's amdgpu.ids complaint), so the
    process keeps a private duplicate of the real stdout for the line and points fd 1 at stderr for everything else."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def _emit(real_fd, obj):
    """Flush libc's and Python's buffers (whatever they hold goes to stderr now), then write the line to the real stdout in one write."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    sys.stderr.flush()
    data = (json.dumps(obj) + "\n").encode()
    while data:
        data = data[os.write(real_fd, data):]


_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch",
              "algorithmic_flops_per_launch", "avg_launch_ms", "episodes_per_launch", "fabric_gbs", "fabric_frac", "executed_tflops", "executed_frac", "executed_f16_tflops", "executed_f16_frac", "path")


def _compact_roof(r):
    """The roofline object of the line: the contract's keys + what they were computed from; notes and the second roof stay in the detail file."""
    if not r:
        return r
    out = {k: r[k] for k in _ROOF_KEYS if k in r}
    if "other_roof" in r:
        out["other_roof"] = {k: r["other_roof"][k] for k in ("bound", "achieved", "peak", "unit", "frac")}
    if "executed_f16_mfma" in r:
        out["executed_f16_mfma_frac"] = r["executed_f16_mfma"]["frac"]
    return out


def _dominant(kernels, roofs):
    if not kernels:
        return None
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    r = roofs.get(dom, {})
    out = {"kernel": dom, "ms": kernels[dom]["ms"], "bound": r.get("bound"), "frac": r.get("frac"), "hbm_frac": round(kernels[dom]["gbs"] / HBM_PEAK_GBS, 4)}
    for key in ("executed_frac", "fabric_frac"):
        if key in r:
            out[key] = r[key]
    return out


def _write_detail(detail):
    """Everything the line used to carry (per-kernel rooflines of every config, notes, per-window baseline counts): gpurun_out/bench_detail.json,
    merged back by gpurun; the judged copy is committed under profiles/r05/."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_detail.json"), "w") as fh:
            json.dump(detail, fh, indent=1)
        return "gpurun_out/bench_detail.json"
    except OSError as e:
        return "not written: %r" % (e,)


def run(args):
    import dkt_amd
    from dkt_amd import ops, distributed

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    real_stdout = _claim_stdout()
    selftest = args.selftest_collective
    backend = "gloo" if selftest else "nccl"
    local = distributed.init_from_env(backend) if world > 1 else 0
    c, s, q, d, desc, n_backbone = CONFIGS[args.config]
    n = c * (s + q)

    if selftest:
        # CPU-only check of the multi-rank plumbing (spawn / rendezvous / flat bucket / timing + attribution protocol); no kernels, no GPU
        dev = torch.device("cpu")
        raw_s = torch.full((c,), float(rank + 1), requires_grad=True)
        mean = torch.zeros(c, requires_grad=True)
        backbone = torch.zeros(n_backbone, requires_grad=True)
        bucket = distributed.GradBucket([backbone, raw_s, mean])
        raw_s.grad = torch.full((c,), float(rank + 1))
        mean.grad = torch.zeros(c)
        backbone.grad = torch.full((n_backbone,), float(rank))
        bucket.allreduce_mean()
        expect = sum(range(1, world + 1)) / world
        ok = bool(torch.allclose(raw_s.grad, torch.full((c,), expect))) and bool(torch.allclose(backbone.grad, torch.full((n_backbone,), (world - 1) / 2.0)))
        # second round on the views the first one attached: no pack copies any more
        backbone.grad.fill_(float(rank))
        bucket.allreduce_mean()
        ok = ok and bucket.copies_last == 0 and bool(torch.allclose(backbone.grad, torch.full((n_backbone,), (world - 1) / 2.0)))
        ar_ms = _allreduce_ms(bucket, world, None)
        if world > 1:
            torch.distributed.destroy_process_group()
        if rank == 0:
            _emit(real_stdout, {"selftest": True, "n_gpus": world, "bucket_floats": bucket.numel, "valid": ok,
                                "collective": _collective_info(bucket, world, ar_ms, backend)})
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the DKT hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dkt_amd._lib.needs_build() and rank == 0:
        dkt_amd._lib.build()
    if world > 1:
        torch.distributed.barrier()

    UNIT_ROWS = os.environ.get("DKT_BENCH_UNIT", "1") != "0"
    traffic = _traffic_table()

    def bucket_fn(st):
        # the flat bucket of the step's one collective: [backbone + bn_out gradient (config size) | hyper-parameter gradients]
        backbone = torch.zeros(st["n_backbone"], device=dev, requires_grad=True)
        bkt = distributed.GradBucket([backbone] + st["params"])
        bkt.attach()
        backbone.grad.fill_(1e-3)
        st["backbone"] = backbone
        return bkt

    b = args.episodes or _default_batch(args.config)
    m = _measure(args.config, b, args, dev, rank, world, bucket_fn, UNIT_ROWS, 3, 1.0, args.steps)
    st, bucket, dt, blocks, total = m["st"], m["bucket"], m["dt"], m["blocks"], m["total"]
    z, raw_s, mean, logp, step = st["z"], st["raw_s"], st["mean"], m["logp"], m["step"]
    ar_ms = _allreduce_ms(bucket, world, dev) if world > 1 else 0.0

    if rank == 0:
        eps = world * b * args.steps / dt
        kernels, roofline_all = _kernel_report(args.config, m, UNIT_ROWS, traffic)
        dom = max(kernels, key=lambda k: kernels[k]["ms"]) if kernels else None
        roofline = roofline_all.get(dom)
        mll_arith = ("factorisations / triangular inverses / K^-1 products as scaled 2-way f16 splits too (16x16x16_f16; diagonal-tile sweeps fp32 on the VALU)"
                     if n + 1 <= 128 else
                     "tile-array / band factorisations on f16-split or fp32 MFMA tiles (DESIGN 4.2b-c; diagonal tiles and the alpha column fp32)")
        arith = ("f32, results fp32-faithful: Gram contractions as a scaled 2-way f16 split of every operand (22 of 24 significand bits, 3 v_mfma_f32_16x16x32_f16, "
                 "fp32 accumulate); " + mll_arith if UNIT_ROWS else
                 "f32: Gram contractions as an exact 3-way bf16 split (6 v_mfma_f32_16x16x32_bf16, fp32 accumulate); " + mll_arith)
        out = {
            "metric": "episodes/sec", "value": round(eps, 1), "unit": "episodes/s", "n_gpus": distributed.world_size(),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32/f16x2-split", "data": "synthetic",
            "config": {"workload": "%s: %s; N=%d D=%d C=%d; training episode fwd+bwd (Gram + %d jittered Cholesky/"
                                   "solve/logdet + MLL + backward)" % (args.config, desc, n, d, c, c),
                       "episodes_per_step_per_gpu": b, "kernel": st["kernel"], "parallelism": "episode-dp%d" % world,
                       "collective": "one all-reduce per step over a flat fp32 bucket of %d floats (%.1f KB: backbone + bn_out of the "
                                     "config + the GP hyper-parameter gradients)" % (bucket.numel, bucket.numel * 4 / 1024.0),
                       "arithmetic": arith},
            "timing": {"blocks": len(blocks), "steps_per_block": args.steps, "block_s_median": round(dt, 6),
                       "block_s_min": round(min(blocks), 6), "block_s_max": round(max(blocks), 6), "timed_s_total": round(total, 4)},
            "valid": m["valid"], "deterministic": m["deterministic"], "roofline": roofline, "roofline_gram_build": roofline_all.get("dkt_gram_f32"),
            "roofline_by_kernel": roofline_all, "kernels": kernels,
            "collective": _collective_info(bucket, world, ar_ms, backend),
        }
    if world == 1 and not args.no_other_configs:
        # every other BASELINE config under the same clock (3 blocks each): value, ms_per_step, per-kernel rooflines
        others = {}
        for cfg in sorted(CONFIGS):
            if cfg == args.config:
                continue
            del m
            torch.cuda.empty_cache()
            bo = _default_batch(cfg)
            so = max(2, min(args.steps, 10 if CONFIGS[cfg][0] * (CONFIGS[cfg][1] + CONFIGS[cfg][2]) <= 128 else 4))
            m = _measure(cfg, bo, args, dev, rank, world, lambda st_: None, UNIT_ROWS, 3, 0.0, so)
            ko, ro = _kernel_report(cfg, m, UNIT_ROWS, traffic)
            so_ = m["st"]
            others[cfg] = {"value": round(bo * so / m["dt"], 1), "unit": "episodes/s", "ms_per_step": round(1e3 * m["dt"] / so, 4),
                           "episodes_per_step": bo, "steps_per_block": so, "blocks": len(m["blocks"]), "valid": m["valid"],
                           "workload": "%s; N=%d D=%d C=%d, kernel %s" % (so_["desc"], so_["n"], so_["d"], so_["c"], so_["kernel"]),
                           "kernels": ko, "roofline_by_kernel": ro}
        out["other_configs"] = others
        del m
        torch.cuda.empty_cache()
        # SURVEY 8d / BASELINE.md 3: the headline config at B = 1 (what the reference's own loop issues, methods/DKT.py:117), 64, 1024 episodes per step -- the GP path
        # only (features resident), forward + backward, under the same clock; launches per step = ABI calls of the step
        sweep = {}
        for bs in (1, 64, 1024):
            ss = 50 if bs <= 64 else 20
            ms_ = _measure(args.config, bs, args, dev, rank, world, lambda st_: None, UNIT_ROWS, 5, 0.0, ss, ktime=False)
            ops.kernel_timing(True)              # the ABI calls of one step, counted apart from the timed blocks
            ms_["step"]()
            torch.cuda.synchronize()
            nl = sum(cnt for cnt, _ in ops.kernel_timing_results().values())
            ops.kernel_timing(False)
            sweep[str(bs)] = {"value": round(bs * ss / ms_["dt"], 1), "ms_per_step": round(1e3 * ms_["dt"] / ss, 4), "launches_per_step": nl, "valid": ms_["valid"]}
            del ms_
            if bs <= 64:
                # the same step captured ONCE into a hipGraph (torch.cuda.CUDAGraph) and replayed: the launch-bound small-batch step without the per-step Python / autograd
                # work (the guide's "capture launch-bound inner loops in hipGraphs"); bitwise equal to the eager step (tests/test_gpu_parity.py)
                gms = _graphed_step_ms(args.config, bs, dev, rank, UNIT_ROWS)
                if gms is not None:
                    sweep[str(bs)]["hipgraph_ms_per_step"] = round(gms, 4)
        out["batch_sweep"] = {"config": args.config, "unit": "episodes/s", "by_episodes_per_step": sweep}
        torch.cuda.empty_cache()
        # the dtype question closed by a number: the same headline step with EXACT fp32 arithmetic everywhere -- Gram forward / backward without the f16 split
        # (DKT_GRAM_SPLIT=0) and the fp32-MFMA marginal likelihood (DKT_MLL_F32MFMA=1) -- from the twins library (same sources, -DDKT_TWINS; a subprocess: the
        # library reads its switches once)
        out["exact_fp32"] = _exact_fp32_run(args)
        out["other_paths_cfg2"] = _aux_paths(dev)
        torch.cuda.empty_cache()
        # the drop-in class' step at the headline's batch (the fused front-end kernels with their rooflines at 8192 episodes)
        out["other_paths_cfg2"]["from_trunk_features_8192"] = _aux_paths(dev, "cfg2", 8192, steps=10, only="from_trunk_features")["from_trunk_features"]
        torch.cuda.empty_cache()
        out["other_paths_cfg4"] = _aux_paths(dev, "cfg4", 512, 64, 5)
        torch.cuda.empty_cache()
        out["other_paths_cfg1"] = _aux_paths(dev, "cfg1", 8192, 2048, 10)          # the Omniglot shape from trunk features: the feature-space episode behind the streaming front end
        torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_test_time and args.config != "cfg0":
            # the forward-only test-time episode (`correct`, DKT.py:199-272) reported separately:
            # condition on the 25 support features, predict the 75 queries (Gram, MLL without gradients, cross Gram, mean + arg-max)
            noise = st["noise"]
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
        if world == 1 and (not args.no_cpu_baseline or args.oracle_check) and args.config != "cfg0":
            import numpy as np
            from oracle import dkt_oracle as O
            nchk = 32                                         # marginal log-likelihood of 32 bench episodes against the float64 oracle
            zc = z[:nchk].detach().cpu()
            sv64 = torch.nn.functional.softplus(raw_s.detach().cpu().double()).numpy()
            hyp = O.GPHypers(sv64, mean.detach().cpu().double().numpy(), np.full(c, 0.1))
            rel = 0.0
            for i in range(nchk):
                ref = O.train_episode(zc[i].double().numpy(), c, hyp)
                rel = max(rel, float(np.abs((logp[i].cpu().numpy() - ref["logp"]) / ref["logp"]).max()))
            out["mll_rel_err"] = rel
            out["mll_rel_err_episodes"] = nchk
        if world == 1 and not args.no_cpu_baseline and args.config != "cfg0":
            zs_cpu = z[:32].detach().cpu()
            res = cpu_baseline(zs_cpu, c, raw_s.detach().cpu(), mean.detach().cpu())
            one = res["1"]["episodes_per_s"]
            allc = res.get("all_cores", {}).get("episodes_per_s")
            facts = res["cores"]
            best, cores = (allc, res["all_cores"]["processes"]) if allc and allc > one else (one, 1)
            out["cpu_baseline"] = {"value": best, "unit": "episodes/s", "cores": cores, "kind": "port",
                                   "sample": "fp32 torch-CPU GPyTorch-structured port of the same training episode (per-class loop, Cholesky, autograd "
                                             "backward), B=1 sequential streams over 32 distinct synthetic Z.  one_thread: 20 warm-up + 200 timed episodes, "
                                             "median of up to 5 repeats (<= 10 s).  all_cores: one pinned single-thread process per effective core "
                                             "(physical cores of the affinity mask, capped by the cgroup CPU quota), episodes all of them completed inside 3 "
                                             "shared 3-s windows, summed.  value = the larger of the two",
                                   "one_thread": one, "all_cores": allc, "cores_effective": facts["cores_effective"],
                                   "parallel_efficiency": res.get("all_cores", {}).get("parallel_efficiency"),
                                   "slowest_process_eps": res.get("all_cores", {}).get("slowest_process_eps"),
                                   "cpu_model": _cpu_model(), "host_cpus": facts["host_cpus"], "affinity_cpus": facts["affinity_cpus"],
                                   "physical_cores": facts["physical_cores_in_affinity"], "cgroup_cpu_quota": facts["cgroup_cpu_quota"],
                                   "by_threads": res}
            if "error" in res.get("all_cores", {}):
                out["cpu_baseline"]["all_cores_error"] = res["all_cores"]["error"]
            out["speedup_vs_cpu"] = round(eps / best, 1)                     # against the BEST host figure (all cores when that is the larger)
            out["speedup_vs_cpu_1thread"] = round(eps / one, 1)
            gp = gpytorch_baseline(zs_cpu, c, raw_s.detach().cpu(), mean.detach().cpu())
            if gp is not None:
                ref0 = O.train_episode(zs_cpu[0].double().numpy(), c, hyp)
                gp["loss_abs_diff_vs_oracle_episode0"] = abs(gp.pop("loss_episode0") - float(ref0["loss"]))
            out["gpytorch_reference"] = gp if gp is not None else "not importable on this box (oracle parity stays unpinned, DESIGN.md section 2)"
        if world == 1 and args.rccl_selftest:
            try:
                out["rccl_selftest"] = _rccl_selftest(dev)
            except Exception as e:  # noqa: BLE001 -- a report, never a reason to lose the bench line
                out["rccl_selftest"] = {"error": repr(e)[:300]}
    if world > 1:
        torch.distributed.destroy_process_group()
    if rank == 0:
        out["detail"] = _write_detail(out)
        _emit(real_stdout, _line_of(out))


def _graphed_step_ms(cfg, b, dev, rank, unit_rows, reps=300):
    """ms per step of the config's training step (Gram -> MLL -> Gram backward + the torch glue around them) replayed from ONE captured hipGraph."""
    try:
        step, _ = _workload(cfg, b, dev, rank, unit_rows)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps
    except Exception as exc:  # noqa: BLE001
        print("hipGraph capture of the %s step at B = %d failed: %s" % (cfg, b, exc), file=sys.stderr)
        return None


def _exact_fp32_run(args):
    """`bench.py --config <headline> --no-other-configs --no-cpu-baseline --no-test-time` in a child with DKT_TWINS=1 DKT_GRAM_SPLIT=0 DKT_MLL_F32MFMA=1: value, ms per step,
    per-kernel ms and the log-likelihood error of the exact-fp32 twin kernels on the same step."""
    import subprocess
    env = dict(os.environ, DKT_TWINS="1", DKT_GRAM_SPLIT="0", DKT_MLL_F32MFMA="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--steps", "10", "--warmup", "2", "--no-other-configs", "--no-cpu-baseline", "--no-test-time", "--oracle-check"]
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
        rec = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
        return {"value": rec["value"], "ms_per_step": rec["ms_per_step"], "kernels_ms": rec.get("kernels_ms"), "mll_rel_err": rec.get("mll_rel_err"), "valid": rec.get("valid"),
                "how": "twins library, DKT_GRAM_SPLIT=0 DKT_MLL_F32MFMA=1"}
    except Exception as exc:  # noqa: BLE001
        return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}


def _line_of(out):
    """The ONE line of the contract (< 8 KB): headline + roofline + roofline_gram_build + cpu_baseline, and one compact record per other config /
    path.  The full per-kernel detail is the `detail` file."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data", "config", "timing", "valid", "deterministic")}
    line["roofline"] = _compact_roof(out["roofline"])
    line["roofline_gram_build"] = _compact_roof(out["roofline_gram_build"])
    line["kernels_ms"] = {k: v["ms"] for k, v in out["kernels"].items() if k not in GLUE_KERNELS}          # (the [B, C]-sized reductions: in the detail file)
    line["collective"] = {k: out["collective"][k] for k in ("bytes", "allreduce_ms", "backend", "ranks", "pack_copies_last_step")}
    if "other_configs" in out:
        line["other_configs"] = {cfg: {"value": o["value"], "ms_per_step": o["ms_per_step"], "episodes_per_step": o["episodes_per_step"], "valid": o["valid"],
                                       "dominant": _dominant(o["kernels"], o["roofline_by_kernel"])} for cfg, o in out["other_configs"].items()}
    for key in ("other_paths_cfg2", "other_paths_cfg4", "other_paths_cfg1"):
        if key in out:
            line[key] = {name: ({"value": o["value"], "ms_per_step": o["ms_per_step"], "valid": o["valid"], "hbm_frac": {k: r["frac"] for k, r in o["roofline"].items()}}
                                if name.endswith("_8192") else          # (compact: kernel times and PMC traffic of this record are in the detail file)
                                {"value": o["value"], "ms_per_step": o["ms_per_step"], "episodes_per_step": o["episodes_per_step"], "valid": o["valid"],
                                "kernels_ms": {k: v for k, v in o["kernels_ms"].items() if k not in GLUE_KERNELS},
                                # per kernel: algorithmic bytes / HIP-event time / 8 TB/s (the full roofline objects are in the detail file)
                                **({"hbm_frac": {k: r["frac"] for k, r in o["roofline"].items()}} if "roofline" in o else {}),
                                **({"traffic_x": {k: r["traffic_over_algorithmic"] for k, r in o["roofline"].items() if "traffic_over_algorithmic" in r}}
                                   if any("traffic_over_algorithmic" in r for r in o.get("roofline", {}).values()) else {})})
                         for name, o in out[key].items()}
    if "test_time_forward" in out:
        line["test_time_forward"] = {k: out["test_time_forward"][k] for k in ("value", "ms_per_step", "episodes_per_step")}
    for k in ("mll_rel_err", "mll_rel_err_episodes", "speedup_vs_cpu", "speedup_vs_cpu_1thread", "gpytorch_reference", "rccl_selftest", "detail", "batch_sweep", "exact_fp32"):
        if k in out:
            line[k] = out[k]
    if "cpu_baseline" in out:
        line["cpu_baseline"] = {k: v for k, v in out["cpu_baseline"].items() if k != "by_threads"}
    return line


def _rccl_selftest(dev):
    """World-1 `nccl` (= RCCL) process group on this GPU: the flat gradient buckets of the two big backbones (ResNet10 + bn_out =
    19.6 MB, ResNet18 = 44.7 MB, + the 2C hyper-parameter gradients and the failure flag) go through GradBucket.allreduce_mean exactly as a
    multi-rank step issues it -- the RCCL launch, the view aliasing (no pack copies), the flag riding in the same collective and the
    `allreduce_ms` plumbing are executed on hardware even when the driver has no multi-GPU node.  NOT a scaling measurement."""
    import socket
    from dkt_amd import distributed
    import torch.distributed as dist
    if dist.is_initialized():
        return {"skipped": "a process group already exists"}
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = {"backend": "nccl (RCCL)", "ranks": 1, "buckets": {}}
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        try:
            res["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
        ok_all = True
        for name, nfl, nh in (("cfg3_resnet10", 4906816, 10), ("cfg4_resnet18", 11177536, 40)):
            backbone = torch.zeros(nfl, device=dev, requires_grad=True)
            hyp = torch.zeros(nh, device=dev, requires_grad=True)
            bkt = distributed.GradBucket([backbone, hyp])
            bkt.attach()
            g = torch.Generator(device=dev).manual_seed(7)
            backbone.grad.copy_(torch.randn(nfl, generator=g, device=dev))
            hyp.grad.fill_(0.25)
            ref = backbone.grad.clone()
            flag = bkt.allreduce_mean(torch.tensor(3.0, device=dev), force=True)
            torch.cuda.synchronize()
            ok = (bool(torch.equal(backbone.grad, ref)) and bkt.copies_last == 0 and float(flag.item()) == 3.0
                  and backbone.grad.data_ptr() == bkt._flat.data_ptr() and float(hyp.grad[0].item()) == 0.25)
            for _ in range(3):
                bkt.allreduce_mean(torch.tensor(0.0, device=dev), force=True)
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s_.record()
            for _ in range(10):
                bkt.allreduce_mean(torch.tensor(0.0, device=dev), force=True)
            e_.record()
            torch.cuda.synchronize()
            res["buckets"][name] = {"bytes": (bkt.numel + 1) * 4, "allreduce_ms": round(s_.elapsed_time(e_) / 10, 4), "pack_copies": bkt.copies_last,
                                    "grads_are_views": backbone.grad.data_ptr() == bkt._flat.data_ptr(), "valid": ok}
            ok_all = ok_all and ok
            del backbone, hyp, bkt, ref
        res["valid"] = ok_all
    finally:
        dist.destroy_process_group()
    return res


def _allreduce_ms(bucket, world, dev, iters=10):
    """The step's collective alone: `iters` all-reduces of the flat bucket bracketed by synchronize (+ barrier), MAX over ranks."""
    if world <= 1 or bucket is None:
        return 0.0
    cuda = dev is not None and dev.type == "cuda"

    def sync():
        if cuda:
            torch.cuda.synchronize()
        torch.distributed.barrier()
        if cuda:
            torch.cuda.synchronize()
    for _ in range(2):
        bucket.allreduce_mean()
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        bucket.allreduce_mean()
    sync()
    t = torch.tensor([(time.perf_counter() - t0) / iters * 1e3], dtype=torch.float64, device=dev if cuda else "cpu")
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def _collective_info(bucket, world, ar_ms, backend):
    """Attribution of the multi-GPU step: what the one collective is, how long it takes alone, what RCCL was told."""
    info = {"op": "all_reduce(SUM) of one flat fp32 bucket, then / world", "bytes": (bucket.numel + 1) * 4 if bucket is not None else 0,
            "allreduce_ms": round(ar_ms, 4), "backend": backend if world > 1 else "none (world 1: no collective is issued)", "ranks": world,
            "NCCL_ALGO": os.environ.get("NCCL_ALGO", "unset (RCCL picks)"), "NCCL_PROTO": os.environ.get("NCCL_PROTO", "unset (RCCL picks)"),
            "pack_copies_last_step": bucket.copies_last if bucket is not None else 0}
    try:
        if world > 1 and backend == "nccl":
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        pass
    return info


def _child(local_rank, args, world, port):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    run(args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--episodes", type=int, default=None,
                    help="episodes per step per GPU (SURVEY.md 8d: B in {1, 64, 1024, 8192}); default 8192, 1024 for the 20-way shapes")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--oracle-check", action="store_true", help="with --no-cpu-baseline: still check 32 bench episodes against the float64 oracle (mll_rel_err)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="only the --config workload (default: the headline config in full, then every other BASELINE config for 3 "
                         "short blocks each, reported under `other_configs`)")
    ap.add_argument("--no-test-time", action="store_true",
                    help="skip the separately reported forward-only test-time episode (profiling runs: keeps the per-kernel "
                         "averages of the trace to the training step's launches)")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="opt-in launch-plumbing check (NOT a measurement): a world-1 RCCL group on this GPU pushes the 19.6 / 44.7 MB gradient buckets "
                         "through GradBucket.allreduce_mean (also: tools/rccl_sanity.py, tests -m gpu)")
    ap.add_argument("--selftest-collective", action="store_true",
                    help="CPU-only (gloo) check of the multi-rank plumbing: spawn, rendezvous, the flat gradient bucket; no kernels")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: one process per GPU, spawned here (the driver's torch.distributed.run path sets WORLD_SIZE itself)
        import socket
        import torch.multiprocessing as mp
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        mp.spawn(_child, args=(args, args.gpus, port), nprocs=args.gpus, join=True)
        return
    run(args)


if __name__ == "__main__":
    main()
