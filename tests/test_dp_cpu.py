"""World-size-2 gloo tests (CPU) of the view-parallel data-parallel helpers (s3gaussian_amd/dp.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from s3gaussian_amd import dp
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    # a parameter set shaped like the real one: row-major tensors, a channels_last plane, and an unused (grad None) head
    params = [torch.nn.Parameter(torch.zeros(1000, 3)), torch.nn.Parameter(torch.zeros(1000, 15, 3)),
              torch.nn.Parameter(torch.zeros(1, 32, 8, 16).contiguous(memory_format=torch.channels_last)),
              torch.nn.Parameter(torch.zeros(64, 64)), torch.nn.Parameter(torch.zeros(7))]
    def fill():
        g = torch.Generator().manual_seed(100 + rank)
        for p in params[:-1]:
            grad = torch.randn(p.shape, generator=g)
            p.grad = grad.contiguous(memory_format=torch.channels_last) if p.dim() == 4 else grad

    fill()
    n0 = dp.GradAllReducer(params, bucket_mb=0.02, inplace_mb=1e-6)()  # every gradient reduced in place (no packing)
    inplace = [p.grad.clone() for p in params[:-1]]
    fill()
    red = dp.GradAllReducer(params, bucket_mb=0.02)  # tiny buckets: exercise the bucket boundaries
    n = red()
    expect = []
    for p in params[:-1]:
        acc = torch.zeros(p.shape)
        for rr in range(world):
            gg = torch.Generator().manual_seed(100 + rr)
            for q_ in params[:-1]:
                t = torch.randn(q_.shape, generator=gg)
                if q_ is p:
                    acc += t
        expect.append(acc / world)
    ok = all(torch.allclose(p.grad, e, atol=1e-6) for p, e in zip(params[:-1], expect)) and params[-1].grad is None
    ok = ok and params[2].grad.is_contiguous(memory_format=torch.channels_last)
    ok = ok and n == sum(p.numel() for p in params[:-1]) and n0 == n
    ok = ok and all(torch.allclose(a, p.grad, atol=1e-6) for a, p in zip(inplace, params[:-1]))
    # densification statistics: the reference's batch semantics (train.py:387-388,435-437) with batch = ranks
    vg = torch.stack([torch.full((10,), float(rank + 1)), torch.full((10,), -2.0 * (rank + 1)), torch.zeros(10)], 1)
    vis = torch.arange(10) % (rank + 2) == 0
    vg = vg * vis[:, None]                       # invisible Gaussians have zero viewspace gradient
    radii = torch.arange(10, dtype=torch.int32) * (rank + 1)
    gsum, any_vis, rmax = dp.reduce_densification_stats(vg, vis, radii)
    exp_g = sum(torch.stack([torch.full((10,), float(rr + 1)), torch.full((10,), -2.0 * (rr + 1))], 1)
                * (torch.arange(10) % (rr + 2) == 0)[:, None] for rr in range(world)) / world
    exp_vis = torch.stack([(torch.arange(10) % (rr + 2) == 0) for rr in range(world)]).any(0)
    ok = ok and torch.allclose(gsum, exp_g, atol=1e-6) and torch.equal(any_vis, exp_vis)
    ok = ok and torch.equal(rmax, torch.arange(10, dtype=torch.int32) * world)
    accum, denom, maxr = torch.zeros(10, 1), torch.zeros(10, 1), torch.zeros(10)
    dp.add_densification_stats(accum, denom, maxr, gsum, any_vis, rmax)
    ok = ok and torch.allclose(accum[:, 0], exp_g.norm(dim=1) * exp_vis, atol=1e-6) and torch.equal(denom[:, 0], exp_vis.float())
    ok = ok and torch.equal(maxr, (torch.arange(10) * world * exp_vis).float())
    # sparse row exchange == dense reduce on per-Gaussian gradients that are zero outside the visible sets
    P = 50
    rows = [torch.nn.Parameter(torch.zeros(P, 15, 3)), torch.nn.Parameter(torch.zeros(P, 1)), torch.nn.Parameter(torch.zeros(P, 4)),
            torch.nn.Parameter(torch.zeros(P, 3))]
    visr = (torch.arange(P) % (3 + rank)) == 0
    gr = torch.Generator().manual_seed(7 + rank)
    for prm in rows[:-1]:
        prm.grad = torch.randn(prm.shape, generator=gr) * visr.view(P, *([1] * (prm.dim() - 1)))
    dense = [prm.grad.clone() for prm in rows[:-1]]
    for d_ in dense:
        dist.all_reduce(d_)
    sx = dp.SparseRowExchange(average=False)
    moved = sx(rows, visr)
    union = sum(((torch.arange(P) % (3 + rr)) == 0).int() for rr in range(world)) > 0
    ok = ok and all(torch.equal(prm.grad, d_) for prm, d_ in zip(rows[:-1], dense)) and rows[-1].grad is None
    ok = ok and sx.last_rows == int(union.sum()) and moved == int(union.sum()) * (45 + 1 + 4)
    # view sharding: disjoint, covering, same permutation on every rank
    mine = dp.shard_views(150, rank, world, seed=0)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    flat = sorted(v for g_ in gathered for v in g_)
    ok = ok and flat == list(range(150))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_view_parallel_helpers_world_size_2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert results == {0: True, 1: True}


def _overlap_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from s3gaussian_amd import dp
    dp.init_from_env(backend="gloo")
    torch.manual_seed(0)   # identical replicas
    big = torch.nn.Parameter(torch.randn(4000, 3))                                                     # reduced by its hook
    plane = torch.nn.Parameter(torch.randn(1, 32, 8, 16).contiguous(memory_format=torch.channels_last))  # hook, NHWC view
    small = torch.nn.Parameter(torch.randn(7))                                                         # bucketed in finish()
    unused = torch.nn.Parameter(torch.randn(5))
    params = [big, plane, small, unused]
    red = dp.OverlappedGradAllReducer(params, bucket_mb=0.001, inplace_mb=0.004)   # threshold 1024 elements
    ok = True
    for step in range(2):   # twice: state must reset between iterations
        for p in params:
            p.grad = None
        x = torch.full((4000, 3), float(rank + 1 + step))            # per-rank data -> per-rank gradients
        loss = (big * x).sum() + (plane * (rank + 2.0)).sum() * 2.0 + (big ** 2).sum() * 0.5 + (small * (rank + 1.0)).sum()
        loss.backward()                                              # big gets two contributions: one accumulate, one hook call
        n = red.finish()
        mean_x = sum(float(r + 1 + step) for r in range(world)) / world
        ok = ok and torch.allclose(big.grad, torch.full((4000, 3), mean_x) + big.detach(), atol=1e-5)
        ok = ok and torch.allclose(plane.grad, torch.full_like(plane, 2.0 * sum(r + 2.0 for r in range(world)) / world), atol=1e-5)
        ok = ok and plane.grad.is_contiguous(memory_format=torch.channels_last)
        ok = ok and torch.allclose(small.grad, torch.full((7,), sum(r + 1.0 for r in range(world)) / world), atol=1e-6)
        ok = ok and unused.grad is None and n == big.numel() + plane.numel() + small.numel()
        ok = ok and not red._inflight and not red._started
    red.remove_hooks()
    # average=False leaves the SUM in place (the Adam kernel applies 1/world: optim.Adam.grad_scale)
    for p in params:
        p.grad = None
    red2 = dp.OverlappedGradAllReducer(params, bucket_mb=0.001, inplace_mb=0.004, average=False)
    ((plane * (rank + 2.0)).sum() + (small * (rank + 1.0)).sum()).backward()
    red2.finish()
    ok = ok and torch.allclose(plane.grad, torch.full_like(plane, sum(r + 2.0 for r in range(world))), atol=1e-5)
    ok = ok and torch.allclose(small.grad, torch.full((7,), sum(r + 1.0 for r in range(world))), atol=1e-6)
    red2.remove_hooks()
    # densify / prune replace nn.Parameter objects (scene/gaussian_model.py:397-494): a reducer built from the OPTIMIZER
    # follows them; the first step after the swap is reduced in finish() (no hook yet), then the hooks are re-armed
    opt_p = torch.nn.Parameter(torch.randn(3000, 3))
    sgd = torch.optim.SGD([{"params": [opt_p], "name": "xyz"}, {"params": [small], "name": "s"}], lr=0.0)
    red3 = dp.OverlappedGradAllReducer(sgd, bucket_mb=0.001, inplace_mb=0.004)
    for step in range(3):
        if step == 1:   # "densification": a new, larger parameter takes the place of the old one in the optimizer group
            opt_p = torch.nn.Parameter(torch.randn(5000, 3))
            sgd.param_groups[0]["params"] = [opt_p]
        for p_ in (opt_p, small):
            p_.grad = None
        ((opt_p * float(rank + 1)).sum() + (small * 2.0).sum()).backward()
        n = red3.finish()
        ok = ok and torch.allclose(opt_p.grad, torch.full_like(opt_p, sum(r + 1.0 for r in range(world)) / world), atol=1e-6)
        ok = ok and n == opt_p.numel() + small.numel()
    ok = ok and red3.rebinds == 1 and set(red3._hooks) == {id(opt_p)}
    red3.remove_hooks()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_overlapped_reducer_world_size_2():
    """Hooks fire during backward (async all-reduce per large gradient), finish() averages + handles the small ones."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert results == {0: True, 1: True}


def _two_phase_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from s3gaussian_amd import dp
    dp.init_from_env(backend="gloo")
    ok = True
    res = {}
    for mode in ("single", "two_phase"):
        torch.manual_seed(0)   # identical replicas
        named = {"xyz": torch.nn.Parameter(torch.randn(3000, 3)), "f_rest": torch.nn.Parameter(torch.randn(3000, 15, 3)),
                 "opacity": torch.nn.Parameter(torch.randn(3000, 1)),
                 "grid": torch.nn.Parameter(torch.randn(1, 32, 8, 16).contiguous(memory_format=torch.channels_last)),
                 "deformation": torch.nn.Parameter(torch.randn(64, 64)), "unused": torch.nn.Parameter(torch.randn(5))}
        opt = torch.optim.Adam([{"params": [p], "lr": 0.01 * (i + 1), "name": n} for i, (n, p) in enumerate(named.items())], lr=0.0, eps=1e-15)
        red = dp.OverlappedGradAllReducer(opt, bucket_mb=0.001, inplace_mb=0.004)
        for step in range(3):
            opt.zero_grad(set_to_none=True)
            k = float(rank + 1 + step)
            loss = sum((p * k).sum() + 0.5 * (p ** 2).sum() for n, p in named.items() if n != "unused")
            loss.backward()
            if mode == "single":
                n_red = red.finish()
                opt.step()
            else:
                n_red = red.finish_and_step(opt)
            ok = ok and n_red == sum(p.numel() for n, p in named.items() if n != "unused")   # every gradient reduced exactly once
            ok = ok and not red._inflight and not red._started
        red.remove_hooks()
        res[mode] = {n: p.detach().clone() for n, p in named.items()}
        ok = ok and all(float(opt.state[p]["step"]) == 3.0 for n, p in named.items() if n != "unused") and len(opt.state[named["unused"]]) == 0
    ok = ok and all(torch.equal(res["single"][n], res["two_phase"][n]) for n in res["single"])
    # step_subset alone: two disjoint subsets == one full step
    torch.manual_seed(1)
    a, b = torch.nn.Parameter(torch.randn(10)), torch.nn.Parameter(torch.randn(4, 3))
    a2, b2 = torch.nn.Parameter(a.detach().clone()), torch.nn.Parameter(b.detach().clone())
    o1 = torch.optim.Adam([{"params": [a]}, {"params": [b]}], lr=0.1)
    o2 = torch.optim.Adam([{"params": [a2]}, {"params": [b2]}], lr=0.1)
    for x, y in ((a, b), (a2, b2)):
        x.grad, y.grad = torch.ones(10) * 0.3, torch.ones(4, 3) * -0.7
    o1.step()
    dp.step_subset(o2, [b2])
    ok = ok and torch.equal(a2.detach(), torch.nn.Parameter(a2.detach()).detach()) and a2.grad is not None and len(o2.state[a2]) == 0
    dp.step_subset(o2, [a2])
    ok = ok and torch.equal(a.detach(), a2.detach()) and torch.equal(b.detach(), b2.detach())
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_two_phase_optimizer_step_equals_the_single_phase_world_size_2():
    """finish_and_step(): early groups stepped while the late collectives are in flight == finish(); optimizer.step()."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_phase_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert results == {0: True, 1: True}


def test_single_process_is_a_noop():
    from s3gaussian_amd import dp
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    assert dp.GradAllReducer([p])() == 0
    assert torch.equal(p.grad, torch.ones(4))
    assert dp.shard_views(10, 0, 1) == dp.shard_views(10, 0, 1)


def test_reduce_skip_flag_is_a_no_op_without_an_asynchronous_forward():
    """dp.reduce_skip_flag() all-reduces the asynchronous rasterizer's overflow flag; a process that has issued no such forward
    (CPU tests, coarse debugging runs with S3G_RASTER_ASYNC=0) has nothing to reduce and must not touch the process group."""
    from s3gaussian_amd import dp, raster_C
    assert not raster_C._async_states
    assert dp.reduce_skip_flag() is None
    assert raster_C.async_status()["calls"] == 0 and raster_C.async_status()["overflows"] == []


def _skip_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from s3gaussian_amd import dp, raster_C
    dp.init_from_env(backend="gloo")
    ok = True
    # rank 0's asynchronous forward "overflowed", rank 1's did not: after finish() / finish_and_step() / the plain reducer BOTH
    # replicas hold 1 in the word their guarded optimizer step reads -- without the caller doing anything
    for mode in ("finish", "finish_and_step", "plain"):
        word = torch.tensor([1 if rank == 0 else 0], dtype=torch.int32)
        seen = []
        raster_C.async_skip_flag = lambda device=None, _w=word: _w
        torch.manual_seed(0)
        p = torch.nn.Parameter(torch.randn(3000, 3))
        small = torch.nn.Parameter(torch.randn(5))

        class Opt(torch.optim.SGD):
            def step(self, closure=None):
                seen.append(int(word.item()))     # what a guarded step would see
                return super().step(closure)

        opt = Opt([{"params": [p], "name": "f_dc"}, {"params": [small], "name": "xyz"}], lr=0.1)
        red = (dp.GradAllReducer(opt) if mode == "plain" else dp.OverlappedGradAllReducer(opt, bucket_mb=0.001, inplace_mb=0.004))
        for it in range(2):
            word.fill_(1 if (rank == 0 and it == 0) else 0)     # second iteration: nobody overflowed
            opt.zero_grad(set_to_none=True)
            ((p * float(rank + 1)).sum() + small.sum()).backward()
            if mode == "finish_and_step":
                red.finish_and_step(opt)
            else:
                red()
                opt.step()
            ok = ok and int(word.item()) == (1 if it == 0 else 0)
        ok = ok and all(s == 1 for s in seen[:len(seen) // 2]) and all(s == 0 for s in seen[len(seen) // 2:]) and len(seen) >= 2
        if hasattr(red, "remove_hooks"):
            red.remove_hooks()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_reducers_agree_on_the_skip_flag_before_any_optimizer_step_world_size_2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_skip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert results == {0: True, 1: True}


def _forced_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", S3G_FORCE_DIST="1")
    from s3gaussian_amd import dp
    r, w, _ = dp.init_from_env(backend="gloo")
    ok = (r, w) == (0, 1) and dist.is_initialized() and dp.active()
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.randn(3000, 3))
    plane = torch.nn.Parameter(torch.randn(1, 32, 8, 16).contiguous(memory_format=torch.channels_last))
    opt = torch.optim.Adam([{"params": [p], "name": "f_dc"}, {"params": [plane], "name": "grid"}], lr=0.01)
    red = dp.OverlappedGradAllReducer(opt, bucket_mb=0.001, inplace_mb=0.004, average=False)
    ok = ok and len(red._hooks) == 2
    ((p ** 2).sum() + (plane * 3.0).sum()).backward()
    want = (p.grad.clone(), plane.grad.clone())
    n = red.finish_and_step(opt)
    ok = ok and n == p.numel() + plane.numel() and torch.equal(p.grad, want[0]) and torch.equal(plane.grad, want[1])
    g, vis, rmax = dp.reduce_densification_stats(torch.ones(6, 3), torch.arange(6) % 2 == 0, torch.arange(6, dtype=torch.int32))
    ok = ok and torch.equal(g, torch.ones(6, 2)) and torch.equal(rmax, torch.arange(6, dtype=torch.int32))
    red.remove_hooks()
    q.put((0, bool(ok)))
    dist.destroy_process_group()


def test_forced_single_rank_group_executes_the_collectives():
    """S3G_FORCE_DIST=1: a process group of ONE rank runs every collective of dp.py (here over gloo; tests/test_rccl_gpu.py is the
    same thing over RCCL on the GPU box)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_forced_worker, args=(_free_port(), q))
    pr.start()
    assert q.get(timeout=120) == (0, True)
    pr.join(timeout=60)


def _stats_skip_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from s3gaussian_amd import dp, raster_C
    dp.init_from_env(backend="gloo")
    ok = True
    res = []
    for it, overflowed_rank in enumerate((0, None, 1)):        # iteration 0: rank 0's forward overflowed; 1: nobody's; 2: rank 1's
        word = torch.tensor([1 if rank == overflowed_rank else 0], dtype=torch.int32)
        raster_C.async_skip_flag = lambda device=None, _w=word: _w
        accum, denom, maxr = torch.zeros(10, 1), torch.zeros(10, 1), torch.zeros(10)
        vis = torch.arange(10) % (rank + 2) == 0
        vg = torch.ones(10, 3) * (rank + 1) * vis[:, None]
        radii = torch.arange(10, dtype=torch.int32) * (rank + 1)
        # bench.py's hook: the statistics are reduced and applied BEFORE the reducer's finish() gets to agree on the word
        g, any_vis, rmax = dp.reduce_densification_stats(vg, vis, radii)
        dp.add_densification_stats(accum, denom, maxr, g, any_vis, rmax)
        ok = ok and int(word.item()) == (0 if overflowed_rank is None else 1)          # agreed by the statistics call itself
        gathered = [None] * world
        dist.all_gather_object(gathered, (accum.tolist(), denom.tolist(), maxr.tolist()))
        ok = ok and gathered[0] == gathered[1]                                          # replicas stay identical
        applied = float(denom.sum()) > 0
        ok = ok and applied == (overflowed_rank is None)
        res.append(applied)
    q.put((rank, bool(ok) and res == [False, True, False]))
    dist.destroy_process_group()


def test_densification_statistics_follow_the_agreed_skip_flag_world_size_2():
    """the guarded bookkeeping must read the AGREED overflow word.  One rank's forward overflows: both ranks skip their
    accumulators (identical denom), although the statistics run before the reducers' finish()."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stats_skip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert results == {0: True, 1: True}


def _replay_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from s3gaussian_amd import dp, pipeline, raster_C
    dp.init_from_env(backend="gloo")
    n = {"fwd": 0}
    state = {"pending": None, "acks": 0}

    def fake_pending(device=None, block=False):
        # rank 1's forward #3 (iteration 4) overflows; ITS host learns of it while issuing iteration 6, rank 0's host never does
        if rank == 1 and state["acks"] == 0 and n["fwd"] >= 6:
            state["pending"] = 3
        return state["pending"]

    def fake_ack(device=None):
        state["pending"], state["acks"] = None, state["acks"] + 1

    raster_C.async_issued = lambda device=None: n["fwd"]
    raster_C.async_replay_pending = fake_pending
    raster_C.async_acknowledge = fake_ack
    raster_C.async_status = lambda device=None, block=False: {}

    class Opt:
        step_calls = 0
        rewound = []

        def rewind_to(self, calls):
            self.rewound.append(self.step_calls - calls)
            self.step_calls = calls

    opt, log = Opt(), []

    def issue(i):
        n["fwd"] += 1
        opt.step_calls += 1

    out = pipeline.run_training_steps(issue, 1, 8, optimizer=opt, log=log)
    gathered = [None] * world
    dist.all_gather_object(gathered, (log, out["rewinds"], Opt.rewound, state["acks"]))
    ok = gathered[0] == gathered[1]                      # both replicas issued the same iterations and rewound to the same one
    ok = ok and log == [1, 2, 3, 4, 5, 6, 4, 5, 6, 7, 8] and out["rewinds"] == [(4, 6)] and Opt.rewound == [3]
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_replay_rewinds_every_replica_to_the_same_iteration_world_size_2():
    """replay under data parallelism.  Only one rank's forward overflows and only that rank's host sees the
    report; dp.agree_min (gloo side group, no device wait) makes both replicas rewind to iteration 4 after issuing iteration 6."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_replay_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert results == {0: True, 1: True}
