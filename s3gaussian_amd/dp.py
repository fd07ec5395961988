"""View-parallel data parallelism over RCCL/xGMI (new functionality: the reference is single-process).

One process per GPU, every rank holds a full replica (Gaussians + HexPlane + MLP) and renders a DIFFERENT
(camera, time) view per step; this is the reference's `batch_size = world_size` semantics (train.py:331-392: mean
loss over the batch, radii max, visibility any, viewspace gradients summed).  Per step:

  * one bucketed all-reduce (average) of every parameter gradient: 59 floats per Gaussian + 35.7 M HexPlane floats
    + the MLP  (~426 MB at 1.2 M Gaussians) in <= `bucket_mb` flat buckets, so RCCL can drive all 7 xGMI links with a
    few large collectives instead of hundreds of small ones;
  * three small all-reduces for the densification statistics (sum of ||viewspace grad||*visible, sum of visible,
    max of radii) so that the replicated densify/prune decisions stay identical on every rank.

Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (tests/test_dp_cpu.py).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """(rank, world_size, local_rank) from torchrun's environment; single-process when WORLD_SIZE is unset/1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("S3G_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_views(num_views: int, rank: int, world: int, seed: int = 0) -> List[int]:
    """Views shuffled with a seed shared by all ranks; rank r takes positions r, r+world, ...."""
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(num_views, generator=g).tolist()
    return perm[rank::world]


def _flat_view(t: torch.Tensor) -> Optional[torch.Tensor]:
    """1-D view sharing storage with t (None if t is neither contiguous nor channels_last)."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(-1)  # a view: NHWC is the physical order
    return None


class GradAllReducer:
    """Bucketed average of .grad over all ranks.  Parameters whose grad is None on this rank (unused heads) are
    skipped; they are None on every rank because all replicas run the same graph."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 256.0, inplace_mb: float = 16.0,
                 average: bool = True):
        self.params = [p for p in params]
        self.average = average   # False: leave the SUM in .grad (the optimizer applies 1/world, optim.Adam.grad_scale)
        self.bucket_elems = int(bucket_mb * 1024 * 1024 / 4)
        self.inplace_elems = int(inplace_mb * 1024 * 1024 / 4)  # gradients at least this big are reduced where they live
        self._buf = None

    def _buckets(self, grads: Sequence[torch.Tensor]):
        cur, n = [], 0
        for g in grads:
            if cur and n + g.numel() > self.bucket_elems:
                yield cur
                cur, n = [], 0
            cur.append(g)
            n += g.numel()
        if cur:
            yield cur

    @torch.no_grad()
    def __call__(self) -> int:
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return 0
        world = dist.get_world_size()
        grads = [p.grad for p in self.params if p.grad is not None]
        total = 0
        small = []
        for g in grads:  # big tensors (xyz/f_rest/planes: ~95 % of the bytes): no pack/unpack copies at all
            v = _flat_view(g) if g.numel() >= self.inplace_elems else None
            if v is None:
                small.append(g)
                continue
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            if self.average:
                v.mul_(1.0 / world)
            total += g.numel()
        for bucket in self._buckets(small):
            n = sum(g.numel() for g in bucket)
            if self._buf is None or self._buf.numel() < n or self._buf.device != bucket[0].device:
                self._buf = torch.empty(max(n, 1), dtype=torch.float32, device=bucket[0].device)
            flat = self._buf[:n]
            off = 0
            for g in bucket:  # pack
                v = _flat_view(g)
                flat[off:off + g.numel()].copy_(v if v is not None else g.contiguous().view(-1))
                off += g.numel()
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            if self.average:
                flat.mul_(1.0 / world)
            off = 0
            for g in bucket:  # unpack in place
                v = _flat_view(g)
                if v is not None:
                    v.copy_(flat[off:off + g.numel()])
                else:
                    g.copy_(flat[off:off + g.numel()].view(g.shape))
                off += g.numel()
            total += n
        return total


class OverlappedGradAllReducer(GradAllReducer):
    """GradAllReducer whose large all-reduces start DURING backward: a post-accumulate hook on every parameter launches
    the collective of that gradient (async, on RCCL's own stream) the moment autograd has finished it, so the wire time
    hides under the rest of the backward pass.  Order of readiness on this path: SH / opacity / scale / rotation
    gradients right after the render glue backward (269 MB at cfg3), MLP weights after the MLP backward, HexPlane planes
    and xyz at the very end.  `finish()` (call between backward and optimizer.step) waits for the collectives in flight,
    averages, and reduces the small gradients in flat buckets like the base class."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 256.0, inplace_mb: float = 1.0,
                 average: bool = True):
        # hook threshold 1 MB: at cfg3 that covers every per-Gaussian array and all but the coarsest planes (~99 % of the
        # bytes, ~35 collectives); the rest (MLP weights, 64x64 planes) goes in one flat bucket in finish()
        super().__init__(params, bucket_mb, inplace_mb, average)
        self._inflight = []   # (work handle, flat view)
        self._started = set()
        self._handles = []
        if dist.is_initialized() and dist.get_world_size() > 1:
            for p in self.params:
                if p.requires_grad and p.numel() >= self.inplace_elems:
                    self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))

    def _on_grad_ready(self, p: torch.nn.Parameter):
        g = p.grad
        v = _flat_view(g) if g is not None else None
        if v is None:
            return
        self._inflight.append((dist.all_reduce(v, op=dist.ReduceOp.SUM, async_op=True), v))
        self._started.add(id(p))

    def remove_hooks(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    @torch.no_grad()
    def finish(self) -> int:
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return 0
        world = dist.get_world_size()
        total = 0
        for work, v in self._inflight:
            work.wait()
            if self.average:
                v.mul_(1.0 / world)
            total += v.numel()
        started, self._inflight, self._started = self._started, [], set()
        # everything the hooks did not cover (small tensors, gradients set outside autograd's accumulation)
        rest = [p for p in self.params if p.grad is not None and id(p) not in started]
        saved, self.params = self.params, rest
        try:
            total += GradAllReducer.__call__(self)
        finally:
            self.params = saved
        return total

    __call__ = finish


@torch.no_grad()
def reduce_densification_stats(viewspace_grad: torch.Tensor, visibility: torch.Tensor, radii: torch.Tensor):
    """Batch semantics of train.py:387-388,435-437 + scene/gaussian_model.py:693-695 across ranks.
    Returns (sum over ranks of ||grad.xy|| on visible Gaussians [P,1], visible count [P,1], max radii [P])."""
    gnorm = torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True) * visibility[:, None].to(viewspace_grad.dtype)
    count = visibility[:, None].to(viewspace_grad.dtype).clone()
    rmax = radii.clone()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(gnorm, op=dist.ReduceOp.SUM)
        dist.all_reduce(count, op=dist.ReduceOp.SUM)
        dist.all_reduce(rmax, op=dist.ReduceOp.MAX)
    return gnorm, count, rmax
