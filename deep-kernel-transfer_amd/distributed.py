"""Episode-parallel data parallelism: one process per GPU, episodes sharded by rank, ONE all-reduce
per optimizer step over a flat fp32 gradient bucket (backbone + bn_out + GP hyper-parameters).

The reference is single-process / single-GPU (no torch.distributed anywhere); this
is the build's addition.  Backend "nccl" is RCCL on ROCm; the CPU tests use "gloo".
Bucket sizes: Conv4 + bn_out(1600) + 10 hypers = 116 298 floats = 465 KB -> latency-bound on xGMI
(one collective, not one per tensor); ResNet18 = 44.7 MB -> ~0.5 ms ring over one 153 GB/s link.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank() -> int:
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend: Optional[str] = None) -> int:
    """torchrun-style bring-up (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    Returns the local rank.  No-op when WORLD_SIZE is unset or 1."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ws <= 1 or (dist.is_available() and dist.is_initialized()):
        return local
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    dist.init_process_group(backend=backend)
    return local


def shard_episodes(n_episodes: int, rank_: Optional[int] = None, world: Optional[int] = None) -> range:
    """Contiguous split of an episode list (test time: 600 episodes) -- rank r gets
    [r*n/W, (r+1)*n/W)."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world is None else world
    lo = (n_episodes * r) // w
    hi = (n_episodes * (r + 1)) // w
    return range(lo, hi)


class GradBucket:
    """Flat fp32 bucket over the parameters that require grad (each tensor once, aliases de-duplicated).
    `allreduce_mean()` averages the gradients over ranks with ONE collective.

    The parameters' `.grad` tensors are VIEWS of the flat buffer: autograd accumulates into an existing `.grad` in place, so a
    backward pass writes straight into the bucket and the collective needs no pack / unpack copies.  A caller that drops the views
    (`p.grad = None`, `optimizer.zero_grad(set_to_none=True)`) is still served: such a gradient is copied in once and the view is
    re-attached after the collective.  `zero_()` clears every gradient without dropping the views."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        self.numel = sum(p.numel() for p in self.params)
        self._flat: Optional[torch.Tensor] = None
        self.copies_last = 0           # gradients that had to be packed by a copy in the last allreduce_mean (0 once the views persist)

    def _buffer(self) -> torch.Tensor:
        p0 = self.params[0]
        if self._flat is None or self._flat.device != p0.device:
            self._flat = torch.zeros(self.numel + 1, device=p0.device, dtype=torch.float32)      # [gradients | flag]
        return self._flat

    def _view(self, p: torch.nn.Parameter, off: int) -> torch.Tensor:
        return self._flat[off:off + p.numel()].view(p.shape)

    def _is_view(self, p: torch.nn.Parameter, off: int) -> bool:
        g = p.grad
        return (g is not None and g.dtype == torch.float32 and g.device == self._flat.device and g.is_contiguous()
                and g.data_ptr() == self._flat.data_ptr() + 4 * off)

    def attach(self) -> None:
        """Make every parameter's .grad a view of the flat buffer (existing gradients are kept)."""
        if not self.params:
            return
        self._buffer()
        off = 0
        for p in self.params:
            if not self._is_view(p, off) and p.dtype == torch.float32:
                v = self._view(p, off)
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
                p.grad = v
            off += p.numel()

    def zero_(self) -> None:
        """Clear every gradient of the bucket's parameters: one fill of the flat buffer for the views, and an in-place zero_() for any
        gradient that is NOT a view of it (a non-fp32 parameter, which attach() leaves alone, or a gradient somebody replaced) -- those
        would otherwise accumulate over the steps, since the training loop no longer calls optimizer.zero_grad() in this mode."""
        if self._flat is not None:
            self._flat.zero_()
        off = 0
        for p in self.params:
            if p.grad is not None and (self._flat is None or not self._is_view(p, off)):
                p.grad.zero_()
            off += p.numel()

    def allreduce_mean(self, flag: Optional[torch.Tensor] = None, force: bool = False) -> Optional[torch.Tensor]:
        """Average the gradients over the ranks.  `flag` (a 0-dim / 1-element tensor, e.g. max |info| of the step) rides in the
        same collective and comes back SUMMED over the ranks, so that every rank learns about a failure on any rank in the
        same step (and raises together instead of leaving the others blocked in the next collective).
        `force`: issue the collective even in a process group of ONE rank (bench.py's RCCL self-test on a 1-GPU box: the same
        code path, launch and view aliasing as a multi-rank step; at world 1 it is otherwise skipped)."""
        forced = force and dist.is_available() and dist.is_initialized()
        if not (is_distributed() or forced) or not self.params:
            return flag
        flat = self._buffer()
        flat[self.numel] = flag.reshape(-1)[0].to(torch.float32) if flag is not None else 0.0
        off, copies = 0, 0
        for p in self.params:
            n = p.numel()
            if not self._is_view(p, off):
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
                    copies += 1
            off += n
        self.copies_last = copies
        red = flat if flag is not None else flat[:self.numel]
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
        out_flag = flat[self.numel].clone() if flag is not None else None
        flat[:self.numel].div_(dist.get_world_size())
        off = 0
        for p in self.params:
            if not self._is_view(p, off):
                if p.dtype == torch.float32:
                    p.grad = self._view(p, off)
                elif p.grad is None:                                   # (non-fp32 parameters cannot alias the fp32 bucket: copied back)
                    p.grad = self._view(p, off).to(p.dtype)
                else:
                    p.grad.copy_(self._view(p, off))
            off += p.numel()
        return out_flag


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if is_distributed():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def gather_accuracies(acc_local: Sequence[float], device=None) -> List[float]:
    """Test time: every rank evaluates its shard of the episode list; all ranks get the full list
    (rank order = episode order for the contiguous split of shard_episodes)."""
    if not is_distributed():
        return list(acc_local)
    w = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    n_local = torch.tensor([len(acc_local)], device=device, dtype=torch.int64)
    sizes = [torch.zeros_like(n_local) for _ in range(w)]
    dist.all_gather(sizes, n_local)
    mx = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(mx, device=device, dtype=torch.float64)
    if len(acc_local):
        buf[:len(acc_local)] = torch.tensor(list(acc_local), device=device, dtype=torch.float64)
    bufs = [torch.zeros_like(buf) for _ in range(w)]
    dist.all_gather(bufs, buf)
    out: List[float] = []
    for s, b in zip(sizes, bufs):
        out += b[:int(s.item())].tolist()
    return out


def broadcast_module_state(module: torch.nn.Module, src: int = 0) -> None:
    """Same initial weights / buffers on every rank."""
    if not is_distributed():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


def average_module_buffers(module: torch.nn.Module) -> None:
    """Mean over ranks of every floating-point buffer (BatchNorm running_mean / running_var), one flat all-reduce; integer
    buffers (num_batches_tracked) take rank 0's value.  No-op outside torch.distributed."""
    if not is_distributed():
        return
    fl = [b for b in module.buffers() if b.is_floating_point()]
    if fl:
        flat = torch.cat([b.detach().reshape(-1).to(torch.float32) for b in fl])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(dist.get_world_size())
        off = 0
        for b in fl:
            n = b.numel()
            b.data.copy_(flat[off:off + n].reshape(b.shape))
            off += n
    for b in module.buffers():
        if not b.is_floating_point():
            dist.broadcast(b.data, src=0)
