"""world_size-2 `gloo` tests of the episode-parallel path (CPU): the flat gradient bucket averages
over ranks with one collective, the averaged gradient equals the mean of the single-episode oracle
gradients (SURVEY.md 8e semantics caveat), episode sharding and accuracy gathering."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import dkt_amd
    from oracle import dkt_oracle as O
    from oracle import dkt_oracle_torch as T
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dkt_amd.distributed.init_from_env("gloo")
    try:
        torch.manual_seed(0)                                  # same init on every rank
        lin = torch.nn.Linear(12, 8).double()
        raw_s = torch.nn.Parameter(torch.zeros(3, dtype=torch.float64))
        mean = torch.nn.Parameter(torch.zeros(3, dtype=torch.float64))
        frozen = torch.nn.Parameter(torch.ones(2, dtype=torch.float64), requires_grad=False)
        params = list(lin.parameters()) + [raw_s, mean, frozen, raw_s]   # alias + frozen must be handled
        bucket = dkt_amd.distributed.GradBucket(params)
        assert bucket.numel == 12 * 8 + 8 + 3 + 3

        def episode_grads(seed):
            for p in params:
                p.grad = None
            x = torch.tensor(np.random.default_rng(seed).standard_normal((12, 12)))
            z = torch.nn.functional.normalize(lin(x), dim=1)
            loss, _, _ = T.classification_loss(z, 3, torch.nn.functional.softplus(raw_s), mean,
                                               torch.full((3,), 0.1, dtype=torch.float64))
            loss.backward()
            return [p.grad.clone() for p in bucket.params]

        mine = episode_grads(100 + rank)                      # rank r draws its own episode
        bucket.allreduce_mean()
        got = [p.grad.clone() for p in bucket.params]
        both = [episode_grads(100 + r) for r in range(world)]
        for i, g in enumerate(got):
            ref = sum(b[i] for b in both) / world
            assert torch.allclose(g, ref, atol=1e-12), (rank, i)
        # sharding + gather
        sh = dkt_amd.distributed.shard_episodes(601)
        accs = [float(e) for e in sh]
        allacc = dkt_amd.distributed.gather_accuracies(accs, device=torch.device("cpu"))
        assert allacc == [float(e) for e in range(601)]
        # failure flag riding in the gradient collective: summed over the ranks, gradients still averaged
        for p_ in bucket.params:
            p_.grad = torch.full_like(p_, float(rank))
        flag = bucket.allreduce_mean(torch.tensor(3.0 if rank == 1 else 0.0))
        assert float(flag) == 3.0 and all(torch.allclose(p_.grad, torch.full_like(p_, (world - 1) / 2.0)) for p_ in bucket.params)
        # the training loop's own step protocol (DKT._zero_grads / _sync_grads, dkt.py): gradients stay VIEWS of the flat bucket over
        # the steps -- cleared by one fill, written by backward, reduced in place, masked in place -- so no step packs by copy
        from dkt_amd.dkt import DKT

        class Host(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.lin = torch.nn.Linear(6, 4)
                self.raw = torch.nn.Parameter(torch.zeros(3))
                self._grad_bucket = None
            _bucket, _zero_grads, _sync_grads = DKT._bucket, DKT._zero_grads, DKT._sync_grads
        torch.manual_seed(1)
        host = Host()
        opt = torch.optim.Adam(host.parameters(), lr=1e-2)
        for step in range(3):
            host._zero_grads(opt)
            xb = torch.tensor(np.random.default_rng(10 * step + rank).standard_normal((5, 6)), dtype=torch.float32)
            loss = (host.lin(xb) ** 2).mean() + (host.raw ** 2).sum() * (1.0 + rank)
            loss.backward()
            mine_w = host.lin.weight.grad.clone()
            bad = host._sync_grads(torch.tensor(0.0))
            assert float(bad) == 0.0
            bkt = host._grad_bucket
            assert bkt.copies_last == 0, (step, bkt.copies_last)
            flat = bkt._flat
            for p_ in bkt.params:
                assert flat.data_ptr() <= p_.grad.data_ptr() < flat.data_ptr() + 4 * flat.numel()
            # the reduced gradient is the mean over the ranks of what each rank's backward wrote
            both_w = [torch.zeros_like(mine_w) for _ in range(world)]
            dist.all_gather(both_w, mine_w)
            assert torch.allclose(host.lin.weight.grad, sum(both_w) / world, atol=1e-6)
            for p_ in host.parameters():
                p_.grad.masked_fill_(torch.tensor(False), 0.0)          # the non-fused failure mask of train_loop: in place
            opt.step()
        # BatchNorm running estimates averaged over the ranks before a checkpoint is written
        bn = torch.nn.BatchNorm1d(4)
        bn.running_mean.fill_(float(rank))
        bn.running_var.fill_(1.0 + 2.0 * rank)
        bn.num_batches_tracked.fill_(10 + rank)
        dkt_amd.distributed.average_module_buffers(bn)
        assert torch.allclose(bn.running_mean, torch.full((4,), (world - 1) / 2.0)) and torch.allclose(bn.running_var, torch.full((4,), float(world)))
        assert int(bn.num_batches_tracked) == 10
        t = dkt_amd.distributed.allreduce_sum_(torch.tensor([1.0 + rank]))
        assert t.item() == sum(1.0 + r for r in range(world))
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gloo_world2_gradient_bucket_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_shard_episodes_partitions():
    import dkt_amd
    for n, w in [(600, 8), (601, 8), (5, 8), (0, 2)]:
        parts = [list(dkt_amd.distributed.shard_episodes(n, r, w)) for r in range(w)]
        assert sum(parts, []) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_bench_spawns_its_own_ranks_and_reduces_the_config_sized_bucket():
    """`bench.py --gpus 2` without a launcher spawns one process per rank itself; the step's one collective carries the
    config's backbone + bn_out bucket plus the 2C hyper-parameter gradients (cfg2: 116 288 + 10 floats, + the failure flag).  CPU / gloo selftest
    of that plumbing -- the kernels themselves need the GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-collective"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    # the driver parses the LAST line of stdout: it must be the JSON record (round 4 lost its bench record to a library banner that followed it)
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert len(res.stdout.strip().splitlines()) == 1, res.stdout
    coll = out.pop("collective")
    assert out == {"selftest": True, "n_gpus": 2, "bucket_floats": 116288 + 10, "valid": True}
    # attribution of the multi-GPU step: the bucket alone is timed, the collective's size / backend / rank count and what RCCL was told
    # are reported, and the second step ran on gradient VIEWS of the flat buffer (no pack copies)
    assert coll["bytes"] == 4 * (116288 + 10 + 1) and coll["ranks"] == 2 and coll["backend"] == "gloo"
    assert coll["allreduce_ms"] > 0.0 and coll["pack_copies_last_step"] == 0 and "NCCL_ALGO" in coll and "NCCL_PROTO" in coll
    # world 1, same contract
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--selftest-collective"], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["selftest"] is True and out["n_gpus"] == 1 and out["valid"] is True


def test_bench_line_is_the_last_and_only_stdout_line_whatever_libraries_print():
    """bench.py keeps a private duplicate of the real stdout for its ONE JSON line and points fd 1 at stderr: C-stdio output a library
    buffers (RCCL's version banner, flushed at exit) and anything the baseline's child processes print can neither follow nor precede the
    line.  Reproduced here with libc puts() before and after the emit, and a child started under _QuietChildren."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = ("import os, sys\n"
             "print('child-out')\n"
             "print('child-err', file=sys.stderr)\n"
             "open(sys.argv[1], 'w').write(os.environ['OMP_NUM_THREADS'])\n")
    parent = ("import ctypes, os, subprocess, sys\n"
              "sys.path.insert(0, sys.argv[1])\n"
              "import bench\n"
              "libc = ctypes.CDLL(None)\n"
              "fd = bench._claim_stdout()\n"
              "libc.puts(b'RCCL version : banner-before')\n"       # sits in libc's buffer (stdout is a pipe) until fflush / exit
              "with bench._QuietChildren():\n"
              "    pr = subprocess.Popen([sys.executable, sys.argv[2], sys.argv[3]])\n"
              "pr.wait()\n"
              "bench._emit(fd, {'metric': 'episodes/sec', 'value': 1.0})\n"
              "libc.puts(b'RCCL version : banner-after')\n")       # flushed at exit, AFTER the line -- to stderr
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        envfile, childfile, parentfile = (os.path.join(td, f) for f in ("omp", "child.py", "parent.py"))
        open(childfile, "w").write(child)
        open(parentfile, "w").write(parent)
        res = subprocess.run([sys.executable, parentfile, root, childfile, envfile], capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, res.stderr[-2000:]
        assert open(envfile).read().strip() == "1"
    assert res.stdout.count("\n") == 1 and json.loads(res.stdout) == {"metric": "episodes/sec", "value": 1.0}
    assert "banner-before" in res.stderr and "banner-after" in res.stderr
    assert "child-out" not in res.stdout + res.stderr and "child-err" not in res.stdout + res.stderr


def test_bench_line_stays_under_8k_with_every_section_filled():
    """The line the driver parses carries the contract's keys + one compact record per other config / path; everything else goes to the detail file."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    roof = dict(kernel="dkt_gram_bwd_f32", bound="hbm", achieved=5000.0, peak=8000.0, unit="GB/s", frac=0.625, traffic=11373158400,
                traffic_unit="x" * 80, traffic_source="profiles/r05/cfg2_summary.txt", algorithmic_bytes_per_launch=11371806720,
                algorithmic_flops_per_launch=289013760000, avg_launch_ms=2.2, episodes_per_launch=8192, other_roof=dict(bound="mfma", achieved=1.0,
                peak=157.3, unit="TFLOP/s", frac=0.01, note="n" * 300), executed_f16_mfma=dict(achieved=1.0, peak=2500.0, unit="TFLOP/s", frac=0.1, note="n" * 200))
    kern = {k: dict(launches=60, ms=1.0, gbs=4000.0, tflops=10.0) for k in ("dkt_gram_f32", "dkt_mll_f32", "dkt_gram_bwd_f32")}
    other = dict(value=1.0e6, unit="episodes/s", ms_per_step=1.0, episodes_per_step=8192, steps_per_block=10, blocks=3, valid=True, workload="w" * 120,
                 kernels=kern, roofline_by_kernel={k: dict(roof, kernel=k) for k in kern})
    path = dict(value=1.0e6, unit="episodes/s", episodes_per_step=2048, ms_per_step=1.6, valid=True, kernels_ms={"dkt_gram_bn_train_f32": 0.4, "dkt_mll_f32": 0.5,
                "dkt_gram_bn_bwd_f32": 0.8}, roofline={k: dict(bound="hbm", achieved=3100.0, peak=8000.0, unit="GB/s", frac=0.39) for k in ("dkt_gram_bn_train_f32", "dkt_gram_bn_bwd_f32")})
    out = dict(metric="episodes/sec", value=2.0e6, unit="episodes/s", n_gpus=1, steps=20, warmup=5, ms_per_step=4.0, higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="f32/f16x2-split", data="synthetic", config=dict(workload="w" * 200, episodes_per_step_per_gpu=8192, kernel="bncossim",
               parallelism="episode-dp1", collective="c" * 160, arithmetic="a" * 500), timing=dict(blocks=12, steps_per_block=20, block_s_median=0.08, block_s_min=0.08,
               block_s_max=0.08, timed_s_total=1.0), valid=True, deterministic=True, roofline=roof, roofline_gram_build=dict(roof, kernel="dkt_gram_f32"),
               roofline_by_kernel={k: roof for k in kern}, kernels=kern, collective=dict(op="o" * 60, bytes=4, allreduce_ms=0.0, backend="none" * 10, ranks=1,
               NCCL_ALGO="unset", NCCL_PROTO="unset", pack_copies_last_step=0), other_configs={c: other for c in ("cfg0", "cfg1", "cfg3", "cfg4", "cfg4_n320")},
               other_paths_cfg2={"from_trunk_features": path, "rbf_per_class_lengthscales": path}, other_paths_cfg4={"from_trunk_features": path,
               "rbf_per_class_lengthscales": path}, other_paths_cfg1={"from_trunk_features": path, "rbf_per_class_lengthscales": path}, test_time_forward=dict(value=1.0, unit="episodes/s", episodes_per_step=4096, ms_per_step=1.0, workload="w" * 100),
               mll_rel_err=9.6e-7, mll_rel_err_episodes=32, speedup_vs_cpu=100.0, speedup_vs_cpu_1thread=13000.0, gpytorch_reference="g" * 90, detail="gpurun_out/bench_detail.json",
               cpu_baseline=dict(value=15000.0, unit="episodes/s", cores=128, kind="port", sample="s" * 600, one_thread=152.0, all_cores=15000.0, cores_effective=128,
               parallel_efficiency=0.8, slowest_process_eps=100.0, cpu_model="AMD EPYC 9575F 64-Core Processor", host_cpus=256, affinity_cpus=256, physical_cores=128,
               cgroup_cpu_quota=None, by_threads={"x": list(range(2000))}))
    line = bench._line_of(out)
    text = json.dumps(line)
    assert len(text) < 8000, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in line
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"]) and "by_threads" not in line["cpu_baseline"]


def test_grad_bucket_zero_clears_gradients_that_are_not_views():
    """GradBucket.zero_() replaces optimizer.zero_grad() in distributed mode: besides the flat fp32 buffer it must clear the gradients that are
    not views of it (attach() leaves non-fp32 parameters alone), or they would accumulate from step to step."""
    import torch
    import dkt_amd
    p32 = torch.nn.Parameter(torch.zeros(5))
    p64 = torch.nn.Parameter(torch.zeros(3, dtype=torch.float64))
    bkt = dkt_amd.distributed.GradBucket([p32, p64])
    bkt.attach()
    (p32.sum() * 2.0 + p64.sum() * 3.0).backward()
    assert p32.grad.data_ptr() == bkt._flat.data_ptr() and p64.grad is not None and float(p64.grad[0]) == 3.0
    bkt.zero_()
    assert float(p32.grad.abs().sum()) == 0.0 and float(p64.grad.abs().sum()) == 0.0
    assert p32.grad.data_ptr() == bkt._flat.data_ptr()          # still a view
    (p32.sum() + p64.sum()).backward()
    assert float(p32.grad[0]) == 1.0 and float(p64.grad[0]) == 1.0


def test_bench_under_the_drivers_launcher_prints_one_json_line():
    """The driver starts N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`:
    rank 0's JSON record must be the only line on the launcher's stdout (every rank points its fd 1 at stderr; only rank 0 writes the record)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-collective"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = res.stdout.strip().splitlines()
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[-1])
    assert out["selftest"] is True and out["n_gpus"] == 2 and out["valid"] is True and out["collective"]["ranks"] == 2
