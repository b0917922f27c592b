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
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    coll = out.pop("collective")
    assert out == {"selftest": True, "n_gpus": 2, "bucket_floats": 116288 + 10, "valid": True}
    # attribution of the multi-GPU step: the bucket alone is timed, the collective's size / backend / rank count and what RCCL was told
    # are reported, and the second step ran on gradient VIEWS of the flat buffer (no pack copies)
    assert coll["bytes"] == 4 * (116288 + 10 + 1) and coll["ranks"] == 2 and coll["backend"] == "gloo"
    assert coll["allreduce_ms"] > 0.0 and coll["pack_copies_last_step"] == 0 and "NCCL_ALGO" in coll and "NCCL_PROTO" in coll
