"""CPU-only: the host-side policy of the asynchronous rasterizer forward (raster_C) -- capacity ladder, capacities derived from
the observed counts -- and the ctypes mirror of its descriptor struct.  The device behaviour is tested in tests/test_async_gpu.py."""
import ctypes

from tests.test_abi_cpu import _struct_fields


def test_async_descriptor_matches_the_header():
    from s3gaussian_amd import _lib
    assert _struct_fields("s3g_raster.h", "s3g_raster_async") == [f[0] for f in _lib.RasterAsync._fields_]
    assert ctypes.sizeof(_lib.RasterAsync) == 4 * 4 + 5 * 8 + 8 + 8 + 8   # + forward_only (int, padded) + sticky_device (round 5) + status_event (round 6)


def test_capacity_ladder_is_monotone_tight_and_repeats():
    from s3gaussian_amd.raster_C import _quantise
    prev = 0
    for n in list(range(1, 5000)) + [10 ** 6 + 7, 2 ** 31 - 5]:
        q = _quantise(n)
        assert q >= n and q <= n * 1.25 + 1 and q >= prev      # never below the request, at most one ladder step above it
        assert _quantise(q) == q                               # a rung maps to itself: sizes repeat from call to call
        prev = q if n < 5000 else 0
    rungs = {_quantise(n) for n in range(1 << 20, 1 << 21, 997)}
    assert len(rungs) <= 5                                     # four steps per octave


def test_capacities_follow_the_largest_counts_seen():
    from s3gaussian_amd import raster_C
    st = object.__new__(raster_C._AsyncState)                  # no device: only the policy
    st.hist = {}
    assert st.caps((1600, 1066)) is None                       # unknown image size: the caller learns the counts first
    st.hist[(1600, 1066)] = [1_460_000, 2_900_000, 900]
    cap_r, cap_s, lds, long_lists = st.caps((1600, 1066))
    assert cap_r >= raster_C._ASYNC_HEADROOM * 1_460_000 and cap_s >= raster_C._ASYNC_HEADROOM * 2_900_000 and cap_s >= cap_r
    assert cap_r >= raster_C._ASYNC_MIN_INSTANCES
    assert lds == 2048 and long_lists == 0                     # twice the longest list, as a power of two
    st.hist[(1600, 1066)][2] = 3000
    assert st.caps((1600, 1066))[2:] == (4096, 1)              # lists may pass 4096: the long-list sort pass is launched too
    st.hist[(64, 64)] = [10, 12, 3]
    cap_r, cap_s, lds, long_lists = st.caps((64, 64))
    assert cap_r == raster_C._quantise(raster_C._ASYNC_MIN_INSTANCES) and cap_s == raster_C._quantise(2 * raster_C._ASYNC_MIN_INSTANCES)
    assert lds == 256 and long_lists == 0
    st.hist[(64, 64)] = [40_000_000, 70_000_000, 3]              # beyond the floor: the capacity follows the counts
    assert st.caps((64, 64))[0] >= 160_000_000


def test_capacities_stay_inside_the_abi_integer_types():
    from s3gaussian_amd import raster_C
    st = object.__new__(raster_C._AsyncState)
    st.hist = {(8, 8): [1_500_000_000, 3_000_000_000, 100_000]}      # absurd counts: the capacities saturate instead of wrapping
    cap_r, cap_s, lds, long_lists = st.caps((8, 8))
    assert cap_r == 0x7fffffff and cap_s == 0xffffffff and lds == 4096 and long_lists == 1


def test_run_training_steps_rewinds_to_the_overflowed_iteration(monkeypatch):
    """Host logic of pipeline.run_training_steps with the rasterizer's status ring mocked: iteration 4 issues the forward that
    overflows (two forwards per iteration here), the report arrives while iteration 6 is being issued, the loop takes three
    iterations of optimizer step counts back, acknowledges, re-issues 4, 5, 6 and carries on; a second report lands only in the
    blocking drain at the end and sends the loop back once more."""
    from s3gaussian_amd import pipeline, raster_C
    issued_calls = {"n": 0}
    # once iteration 7 has issued its forwards (#12, #13), forward #6 is reported (iteration 4 issued forwards #6 and #7)
    state = {"pending": None, "acks": 0, "late": False}

    def fake_issued(device=None):
        return issued_calls["n"]

    def fake_pending(device=None, block=False):
        if state["pending"] is None and issued_calls["n"] >= 14 and state["acks"] == 0:
            state["pending"] = 6
        if block and state["acks"] == 1 and not state["late"]:
            state["late"], state["pending"] = True, issued_calls["n"] - 2      # the first forward of the last iteration
        return state["pending"]

    def fake_ack(device=None):
        state["pending"], state["acks"] = None, state["acks"] + 1

    monkeypatch.setattr(raster_C, "async_issued", fake_issued)
    monkeypatch.setattr(raster_C, "async_replay_pending", fake_pending)
    monkeypatch.setattr(raster_C, "async_acknowledge", fake_ack)

    class Opt:
        """stands for optim.Adam: one step() per iteration, journalled (rewind_to takes launches back by COUNT of step() calls)."""
        step_calls = 0
        rewound = []

        def rewind_to(self, calls):
            self.rewound.append(self.step_calls - calls)
            self.step_calls = calls

    log = []
    opt = Opt()

    def issue(i):
        issued_calls["n"] += 2
        opt.step_calls += 1

    monkeypatch.setattr(raster_C, "async_status", lambda device=None, block=False: {})
    out = pipeline.run_training_steps(issue, 1, 8, optimizer=opt, log=log)
    # forwards: it1 = #0,1  it2 = #2,3  it3 = #4,5  it4 = #6,7 ...  it7 = #12,13
    assert log[:7] == [1, 2, 3, 4, 5, 6, 7] and log[7:12] == [4, 5, 6, 7, 8] and log[12:] == [8]
    assert out["rewinds"] == [(4, 7), (8, 8)] and Opt.rewound == [4, 1] and out["issued"] == len(log) == 13
    assert raster_C.REPLAY is False and pipeline._replay_loop is None


def test_surgery_barrier_hands_control_back_before_a_host_side_mutation(monkeypatch):
    """a densify / prune / opacity reset inside issue(i) is a HOST-side mutation the device's freeze does not cover.  With
    pipeline.surgery_barrier() in front of it, an overflow reported while iteration 5 is about to mutate sends the loop back to the
    overflowed iteration 3 WITHOUT the mutation having run; it runs once, when iteration 5 is issued again on a thawed model.  The
    optimizer had launched steps for iterations 1-4 (not 5: the barrier sits before its step): exactly 3 and 4 are taken back."""
    from s3gaussian_amd import pipeline, raster_C
    n = {"fwd": 0}
    state = {"pending": None, "acks": 0}

    def fake_pending(device=None, block=False):
        if block and state["acks"] == 0 and n["fwd"] >= 5:
            state["pending"] = 2                       # forward #2 = iteration 3
        return state["pending"]

    def fake_ack(device=None):
        state["pending"], state["acks"] = None, state["acks"] + 1

    monkeypatch.setattr(raster_C, "async_issued", lambda device=None: n["fwd"])
    monkeypatch.setattr(raster_C, "async_replay_pending", fake_pending)
    monkeypatch.setattr(raster_C, "async_acknowledge", fake_ack)
    monkeypatch.setattr(raster_C, "async_status", lambda device=None, block=False: {})

    class Opt:
        step_calls = 0
        rewound = []

        def rewind_to(self, calls):
            self.rewound.append(self.step_calls - calls)
            self.step_calls = calls

    opt, mutations, log = Opt(), [], []

    def issue(i):
        n["fwd"] += 1                                  # forward + backward of iteration i
        if i == 5:
            pipeline.surgery_barrier()                 # train.py:489-516 territory: densify / prune / reset follow
            mutations.append(i)
        opt.step_calls += 1

    out = pipeline.run_training_steps(issue, 1, 6, optimizer=opt, log=log)
    assert log == [1, 2, 3, 4, 3, 4, 5, 6] and mutations == [5] and out["rewinds"] == [(3, 5)] and Opt.rewound == [2]
    pipeline.surgery_barrier()                          # outside a replay loop: no-op


def test_adam_journal_takes_back_exactly_what_the_dropped_launches_advanced():
    """optim.Adam.rewind_to (host logic, no launch): step counts are taken back per PARAMETER -- a parameter that had no gradient in a
    dropped iteration, or was stepped by the other launch of a two-phase step, keeps its count (rewind(n) subtracted n
    from every state)."""
    import torch
    from s3gaussian_amd import optim
    a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(3))
    opt = optim.Adam([{"params": [a]}, {"params": [b]}], lr=0.1)
    sa, sb = opt.state[a], opt.state[b]
    sa["step"], sb["step"] = torch.tensor(0.0), torch.tensor(0.0)

    def launched(states):              # what step() records once its launches are out
        for st in states:
            st["step"] += 1
        opt.step_calls += 1
        opt._journal.append((opt.step_calls, list(states)))

    launched([sa, sb])                 # iteration 1
    mark = opt.step_calls
    launched([sa])                     # iteration 2: b had no gradient
    launched([sa])                     # iteration 3, phase 1 of a two-phase step
    launched([sb])                     # iteration 3, phase 2
    assert (float(sa["step"]), float(sb["step"]), opt.step_calls) == (3.0, 2.0, 4)
    assert opt.rewind_to(mark) == 3 and (float(sa["step"]), float(sb["step"]), opt.step_calls) == (1.0, 1.0, 1)
    launched([sa, sb])
    opt.rewind(1)
    assert (float(sa["step"]), float(sb["step"]), opt.step_calls) == (1.0, 1.0, 1)
