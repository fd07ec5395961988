"""CPU-only: the host-side policy of the asynchronous rasterizer forward (raster_C) -- capacity ladder, capacities derived from
the observed counts -- and the ctypes mirror of its descriptor struct.  The device behaviour is tested in tests/test_async_gpu.py."""
import ctypes

from tests.test_abi_cpu import _struct_fields


def test_async_descriptor_matches_the_header():
    from s3gaussian_amd import _lib
    assert _struct_fields("s3g_raster.h", "s3g_raster_async") == [f[0] for f in _lib.RasterAsync._fields_]
    assert ctypes.sizeof(_lib.RasterAsync) == 4 * 4 + 5 * 8 + 8   # + forward_only (int) and its tail padding


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
