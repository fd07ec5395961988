"""View-parallel data parallelism over RCCL/xGMI (new functionality: the reference is single-process).

One process per GPU, every rank holds a full replica (Gaussians + HexPlane + MLP) and renders a DIFFERENT
(camera, time) view per step; this is the reference's `batch_size = world_size` semantics (train.py:331-392: mean
loss over the batch, radii max, visibility any, viewspace gradients summed).  Per step:

  * one bucketed all-reduce (average) of every parameter gradient: 59 floats per Gaussian + 35.7 M HexPlane floats
    + the MLP  (~426 MB at 1.2 M Gaussians) in <= `bucket_mb` flat buckets, so RCCL can drive all 7 xGMI links with a
    few large collectives instead of hundreds of small ones;
  * three small all-reduces for the densification statistics (viewspace gradient vectors summed and scaled by
    1/world before the norm, visibility ANY, radii MAX -- exactly train.py:387-388,435-437 at batch_size = world)
    so that the replicated densify/prune decisions stay identical on every rank;
  * optionally (`SparseRowExchange`) the per-Gaussian gradients that are exactly zero outside the union of the ranks'
    visible sets (SH, opacity, scale, rotation: 56 of the 59 floats per Gaussian) travel as compact visible rows.

Densify / prune replace the per-Gaussian nn.Parameters (scene/gaussian_model.py:397-494): build the reducers from the
OPTIMIZER (parameters are then resolved from `optimizer.param_groups` at every reduce) or call `rebind()` afterwards.
No multi-GPU scaling curve has been measured on hardware yet (gpurun leases one GPU); see DESIGN.md section 8.

Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (tests/test_dp_cpu.py).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def force_dist() -> bool:
    """S3G_FORCE_DIST=1: run every collective of this module even at world size 1 (a process group of ONE rank).  That is how the
    RCCL code path -- library load, communicator init, in-place all-reduces on channels_last views, the post-accumulate hooks,
    stream ordering against the optimizer kernel -- is executed on a one-GPU lease (tests/test_rccl_gpu.py, `bench.py --gpus 1`
    under the flag); the results equal the non-distributed step bit for bit because a sum over one rank is the identity."""
    return os.environ.get("S3G_FORCE_DIST", "0") == "1"


def active() -> bool:
    """True when the collectives below should run: an initialised process group of more than one rank, or of one rank under
    S3G_FORCE_DIST=1."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or force_dist())


def init_from_env(backend: Optional[str] = None) -> tuple:
    """(rank, world_size, local_rank) from torchrun's environment; single-process when WORLD_SIZE is unset/1 (unless
    S3G_FORCE_DIST=1 asks for a process group of one rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_dist()) and not dist.is_initialized():
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


def _resolve(source) -> List[torch.nn.Parameter]:
    """Current parameter list of `source`: an optimizer (param_groups, the object the reference's densification code
    edits), a callable returning parameters, or a plain iterable captured once."""
    if isinstance(source, torch.optim.Optimizer):
        return [p for g in source.param_groups for p in g["params"]]
    if callable(source):
        return list(source())
    return source


class GradAllReducer:
    """Bucketed average of .grad over all ranks.  Parameters whose grad is None on this rank (unused heads) are
    skipped; they are None on every rank because all replicas run the same graph.

    `params`: a torch optimizer or a zero-argument callable (resolved at EVERY reduce, so parameters replaced by
    densify / prune are picked up), or a fixed iterable of parameters (then call `rebind(new_params)` after any
    operation that replaces nn.Parameter objects -- a stale list would silently stop reducing those gradients)."""

    def __init__(self, params, bucket_mb: float = 256.0, inplace_mb: float = 16.0, average: bool = True):
        self._source = params if (isinstance(params, torch.optim.Optimizer) or callable(params)) else [p for p in params]
        self.average = average   # False: leave the SUM in .grad (the optimizer applies 1/world, optim.Adam.grad_scale)
        self.bucket_elems = int(bucket_mb * 1024 * 1024 / 4)
        self.inplace_elems = int(inplace_mb * 1024 * 1024 / 4)  # gradients at least this big are reduced where they live
        self._buf = None
        self._only = None   # set by subclasses to restrict one call to a subset

    @property
    def params(self) -> List[torch.nn.Parameter]:
        return _resolve(self._source)

    def rebind(self, params=None) -> None:
        """Adopt a new parameter source (same kinds as the constructor); no argument = re-resolve the current one."""
        if params is not None:
            self._source = params if (isinstance(params, torch.optim.Optimizer) or callable(params)) else [p for p in params]

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
        if not active():
            return 0
        if self._only is None:
            reduce_skip_flag()       # replicas drop the same optimizer steps (see OverlappedGradAllReducer._agree_on_skip)
        world = dist.get_world_size()
        grads = [p.grad for p in (self._only if self._only is not None else self.params) if p.grad is not None]
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

    def __init__(self, params, bucket_mb: float = 256.0, inplace_mb: float = 1.0, average: bool = True):
        # hook threshold 1 MB: at cfg3 that covers every per-Gaussian array and all but the coarsest planes (~99 % of the
        # bytes, ~35 collectives); the rest (MLP weights, 64x64 planes) goes in one flat bucket in finish()
        super().__init__(params, bucket_mb, inplace_mb, average)
        self._inflight = []   # (work handle, flat view)
        self._skip_agreed = False
        self._started = set()
        self._hooks = {}      # id(param) -> (param, hook handle)
        self.rebinds = 0      # how often the hooked set had to follow replaced parameters (diagnostics / tests)
        self.rebind()

    def rebind(self, params=None) -> None:
        """(Re-)registers the post-accumulate hooks on the CURRENT parameters; stale hooks (parameters densify / prune
        replaced) are removed.  finish() calls this itself when it sees the optimizer's parameters changed, so a missed
        rebind costs one iteration of non-overlapped reduction, never a silent divergence."""
        super().rebind(params)
        if not active():
            return
        want = {id(p): p for p in self.params if p.requires_grad and p.numel() >= self.inplace_elems}
        for k in [k for k in self._hooks if k not in want]:
            self._hooks.pop(k)[1].remove()
        for k, p in want.items():
            if k not in self._hooks:
                self._hooks[k] = (p, p.register_post_accumulate_grad_hook(self._on_grad_ready))

    def _on_grad_ready(self, p: torch.nn.Parameter):
        g = p.grad
        v = _flat_view(g) if g is not None else None
        if v is None:
            return
        self._inflight.append((dist.all_reduce(v, op=dist.ReduceOp.SUM, async_op=True), v))
        self._started.add(id(p))

    def remove_hooks(self):
        for _, h in self._hooks.values():
            h.remove()
        self._hooks = {}

    def _agree_on_skip(self) -> None:
        """Every replica must drop the SAME optimizer steps: the overflow word of the host-asynchronous rasterizer forward is
        all-reduced (MAX, 4 bytes, no host wait) before the first optimizer kernel of the iteration can read it.  Called once per
        iteration by finish() / finish_and_step() -- a data-parallel user of this class cannot forget it (only
        bench.py's hook used to call dp.reduce_skip_flag, so one replica could skip a step the others applied)."""
        if self._skip_agreed:
            return
        self._skip_agreed = True
        reduce_skip_flag()

    @torch.no_grad()
    def finish(self) -> int:
        if not active():
            return 0
        self._agree_on_skip()
        self._skip_agreed = False            # the next iteration agrees anew
        world = dist.get_world_size()
        total = 0
        for work, v in self._inflight:
            work.wait()
            if self.average:
                v.mul_(1.0 / world)
            total += v.numel()
        started, self._inflight, self._started = self._started, [], set()
        current = self.params
        # everything the hooks did not cover: small tensors, gradients set outside autograd's accumulation, and parameters
        # that replaced hooked ones since the last rebind (their gradients exist but no hook fired)
        rest = [p for p in current if p.grad is not None and id(p) not in started]
        self._only = rest
        try:
            total += GradAllReducer.__call__(self)
        finally:
            self._only = None
        hookable = {id(p) for p in current if p.requires_grad and p.numel() >= self.inplace_elems}
        if hookable != set(self._hooks):
            self.rebinds += 1
            self.rebind()
        return total

    __call__ = finish

    @torch.no_grad()
    def finish_and_step(self, optimizer: torch.optim.Optimizer, late_groups: Sequence[str] = ("grid", "xyz", "deformation")) -> int:
        """finish() + optimizer.step() in TWO PHASES, so that the optimizer does not sit idle behind the last collectives.
        The gradients become ready in a fixed order on this path -- SH / opacity / scale / rotation right after the render
        glue backward, the HexPlane planes (143 MB) and xyz at the very end -- so when backward returns the early collectives
        are done or nearly done while ~157 MB are still on the wire.  Phase 1 waits only for the collectives of the groups NOT
        named in `late_groups` (the stream then depends on nothing later: RCCL executes them in issue order) and steps exactly
        those parameters; phase 2 waits for the rest, reduces what no hook covered, and steps the remaining parameters.  Every
        parameter is stepped exactly once with its fully reduced gradient: the result equals finish(); optimizer.step()
        (tests/test_dp_cpu.py).  Group names are the reference's (scene/gaussian_model.py:177-187)."""
        if not active():
            optimizer.step()
            return 0
        self._agree_on_skip()                # before the FIRST of the two optimizer launches reads the word
        world = dist.get_world_size()
        late_ids = {id(p) for g in optimizer.param_groups if g.get("name") in late_groups for p in g["params"]}
        early_params = [p for p in self.params if p.grad is not None and id(p) in self._started and id(p) not in late_ids]
        early_ptrs = {(_flat_view(p.grad).data_ptr() if _flat_view(p.grad) is not None else None) for p in early_params}
        total, rest_works = 0, []
        for work, v in self._inflight:
            if v.data_ptr() in early_ptrs:
                work.wait()
                if self.average:
                    v.mul_(1.0 / world)
                total += v.numel()
            else:
                rest_works.append((work, v))
        step_subset(optimizer, early_params)                      # runs while the late collectives are still in flight
        self._inflight = rest_works          # (the early parameters stay in _started: finish() must not reduce them again)
        stepped = {id(p) for p in early_params}
        total += self.finish()
        step_subset(optimizer, [p for p in self.params if id(p) not in stepped])
        return total


@torch.no_grad()
def step_subset(optimizer: torch.optim.Optimizer, params: Sequence[torch.nn.Parameter]) -> None:
    """optimizer.step() restricted to `params`: every torch optimizer (and s3gaussian_amd.optim.Adam) skips parameters whose
    .grad is None, so the other parameters' gradients are hidden for the duration of the call.  Per-parameter state (Adam's step
    count and moments) only advances for the parameters stepped, so stepping two disjoint subsets equals one full step."""
    keep = {id(p) for p in params}
    hidden = []
    for g in optimizer.param_groups:
        for p in g["params"]:
            if id(p) not in keep and p.grad is not None:
                hidden.append((p, p.grad))
                p.grad = None
    try:
        if any(p.grad is not None for p in params):
            optimizer.step()
    finally:
        for p, gr in hidden:
            p.grad = gr


@torch.no_grad()
def reduce_densification_stats(viewspace_grad: torch.Tensor, visibility: torch.Tensor, radii: torch.Tensor,
                               grads_are_sums: bool = True):
    """The reference's batch semantics (train.py:387-388 radii max / visibility any, :435-437 viewspace gradients SUMMED
    over the views of a batch whose loss is the batch MEAN) with batch = the ranks' views.

    viewspace_grad [P,>=2]: this rank's d(loss_rank)/d(means2D) of its own un-averaged loss (grads_are_sums=True, what
    training_step produces; the 1/world of the batch mean is applied here) or already scaled by 1/world (False).
    Returns (viewspace gradient of the batch [P,2], visible in ANY view [P] bool, max radii [P]); feed them to
    `add_densification_stats` (scene/gaussian_model.py:693-695: accum += ||grad.xy||, denom += 1 where visible)."""
    g = viewspace_grad[:, :2].contiguous().clone()
    any_vis = visibility.to(torch.int32).clone()
    rmax = radii.clone()
    if active():
        # the bookkeeping that consumes these statistics is guarded by the overflow word: agree on it BEFORE it is read
        reduce_skip_flag(viewspace_grad.device if viewspace_grad.is_cuda else None)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        dist.all_reduce(any_vis, op=dist.ReduceOp.MAX)
        dist.all_reduce(rmax, op=dist.ReduceOp.MAX)
        if grads_are_sums:
            g.mul_(1.0 / dist.get_world_size())
    return g, any_vis.bool(), rmax


_skip_agreed_for = {}     # device index -> count of training forwards issued when the word was last all-reduced


@torch.no_grad()
def reduce_skip_flag(device=None):
    """Host-asynchronous rasterizer (raster_C.ASYNC): a forward whose instance count exceeded its speculative arena renders
    nothing, back-propagates zeros, and flags that in a device word which the guarded Adam step and the guarded densification
    bookkeeping honour (include/s3g_optim.h::s3g_adam_step_guarded, s3g_densify_stats_guarded).  Replicas must drop the SAME
    steps, so the flag is all-reduced (MAX, one 4-byte collective, no host wait) IN PLACE: every guarded consumer reads the same
    word.  -> the int32 [1] device tensor, or None when this rank has issued no asynchronous forward (then nothing is reduced:
    every rank runs the same code path, so all or none have one).
    Idempotent per training forward: the first guarded consumer of an iteration -- dp.reduce_densification_stats in bench.py's
    hook, else the reducers' finish() / finish_and_step() -- triggers the one collective, later calls of the same iteration find
    the word already agreed (the statistics used to read the rank-LOCAL word, because only the reducers, which run after
    the hook, agreed on it: a rank whose forward overflowed skipped its accumulators while the others applied theirs)."""
    from . import raster_C
    flag = raster_C.async_skip_flag(device)
    if flag is None:
        return None
    if active():
        st = raster_C._state_of(device)
        if st is not None:      # (None: the word is a test double -- reduce every time)
            if _skip_agreed_for.get(st.device.index) == st.train_forwards:
                return flag
            _skip_agreed_for[st.device.index] = st.train_forwards
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)     # in place: the view into the status ring now holds the batch's verdict
    return flag


_host_group = None


def agree_min(value: Optional[int]) -> Optional[int]:
    """Smallest `value` over all ranks (None = "nothing to report"; None if every rank says so), agreed on the HOST: one 8-byte
    all-reduce over a gloo side group, which waits for the other ranks' hosts, never for a device.  pipeline.run_training_steps
    uses it under data parallelism so that every replica rewinds to the same iteration at the same point of its loop (a collective
    on the device stream would make every host wait for its GPU once per iteration)."""
    global _host_group
    if not active():
        return value
    if _host_group is None:
        _host_group = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
    big = (1 << 62)
    t = torch.tensor([big if value is None else int(value)], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=_host_group)
    v = int(t.item())
    return None if v >= big else v


@torch.no_grad()
def add_densification_stats(xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor,
                            viewspace_grad_xy: torch.Tensor, visible: torch.Tensor, radii: torch.Tensor) -> None:
    """train.py:489-493 + scene/gaussian_model.py:693-695 on the reduced statistics (one fused HIP pass on the GPU:
    include/s3g_optim.h::s3g_densify_stats; plain torch on CPU tensors for the gloo tests)."""
    if xyz_gradient_accum.is_cuda:
        from .optim import densify_stats
        densify_stats(xyz_gradient_accum, denom, max_radii2D, viewspace_grad_xy, radii, visible)
        return
    from . import raster_C
    flag = raster_C.async_skip_flag()            # CPU tensors (gloo tests): the same guard, read on the host
    if flag is not None and int(flag.item()) != 0:
        return
    xyz_gradient_accum[visible] += torch.norm(viewspace_grad_xy[visible, :2], dim=-1, keepdim=True)
    denom[visible] += 1
    max_radii2D[visible] = torch.max(max_radii2D[visible], radii[visible].to(max_radii2D.dtype))


class SparseRowExchange:
    """Optional sparse gradient exchange for per-Gaussian arrays whose rows are EXACTLY zero for Gaussians no rank sees:
    d/d(SH dc, SH rest, opacity, scaling, rotation) -- 56 of the 59 floats per Gaussian (xyz also receives the HexPlane
    gradient of the dx / dshs regularisers and stays dense).  Per step: union of the ranks' visibility masks (one
    all-reduce MAX over P bytes), compact the union's rows of every listed gradient into one flat buffer, ONE all-reduce,
    scatter back.  At cfg3 one view sees ~22 % of the Gaussians, so for small world sizes this moves a fraction of the
    269 MB the dense path moves; with many ranks the union approaches P and the dense path is as good.  Opt-in; results
    are identical to the dense reduce (tests/test_dp_cpu.py), rows outside the union are never touched."""

    def __init__(self, average: bool = True):
        self.average = average
        self.last_rows = 0

    @torch.no_grad()
    def __call__(self, params: Sequence[torch.nn.Parameter], visibility: torch.Tensor) -> int:
        if not active():
            return 0
        world = dist.get_world_size()
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return 0
        P = visibility.shape[0]
        assert all(g.shape[0] == P and g.is_contiguous() for g in grads), "per-Gaussian row-major gradients expected"
        union = visibility.to(torch.uint8).clone()
        dist.all_reduce(union, op=dist.ReduceOp.MAX)
        idx = union.nonzero(as_tuple=True)[0]           # identical on every rank
        self.last_rows = int(idx.numel())
        widths = [g[0].numel() for g in grads]
        flat = torch.cat([g.view(P, -1).index_select(0, idx) for g in grads], dim=1)   # [rows, sum(widths)]
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if self.average:
            flat.mul_(1.0 / world)
        off = 0
        for g, w in zip(grads, widths):
            g.view(P, -1).index_copy_(0, idx, flat[:, off:off + w])
            off += w
        return int(flat.numel())
