"""hexplane.hip shares a bilinear tap between lanes as (nw key, ix - x0, iy - y0) and recomputes x1 - ix as 1 - (ix - x0).
Claim in the source: bit-identical to torch's grid_sampler weights, because ix - x0 is exact in fp32 (Sterbenz; x0 = 0
trivially) and both expressions are then the correct rounding of the same real number.  Checked here in numpy fp32 on the
whole coordinate range the sampler produces (align_corners unnormalisation, border clip), incl. every representable value
near the integers of small grids."""
import numpy as np
import pytest


def weights_reference(ix):
    x0 = np.floor(ix)
    x1 = x0 + np.float32(1)
    return (x1 - ix).astype(np.float32), (ix - x0).astype(np.float32)


def weights_shared(ix):
    x0 = np.floor(ix)
    fx = (ix - x0).astype(np.float32)
    return (np.float32(1) - fx).astype(np.float32), fx


@pytest.mark.parametrize("W", [2, 25, 64, 128, 512, 4096])
def test_one_minus_fraction_is_bit_identical(W):
    rng = np.random.default_rng(W)
    u = rng.uniform(-1.3, 1.3, 2_000_000).astype(np.float32)
    ix = ((u + np.float32(1)) / np.float32(2)) * np.float32(W - 1)           # grid_sampler_unnormalize, align_corners
    ix = np.minimum(np.float32(W - 1), np.maximum(ix, np.float32(0)))        # clip_coordinates
    # plus the neighbourhoods of every integer of the grid (both sides, a few ulps)
    ints = np.arange(0, min(W, 300), dtype=np.float32)
    near = np.concatenate([np.nextafter(ints, np.float32(np.inf)), np.nextafter(ints, np.float32(-np.inf)), ints,
                           ints + np.float32(0.5), ints + np.float32(1e-7), ints + np.float32(0.99999994)])
    ix = np.concatenate([ix, np.clip(near, 0, W - 1).astype(np.float32)])
    g_ref, f_ref = weights_reference(ix)
    g_new, f_new = weights_shared(ix)
    assert np.array_equal(f_ref.view(np.uint32), f_new.view(np.uint32))
    assert np.array_equal(g_ref.view(np.uint32), g_new.view(np.uint32))


def test_packed_tap_word_round_trips():
    """key << 4 | flags in one 32-bit word: keys up to 2^24 - 1 (check_desc rejects larger planes), four flag bits."""
    keys = np.array([0, 1, 12345, 2**24 - 1], dtype=np.uint32)
    for flags in range(16):
        word = (keys << np.uint32(4)) | np.uint32(flags)
        assert np.array_equal(word >> np.uint32(4), keys) and np.all((word & np.uint32(15)) == flags)
    assert (np.uint64(2**24 - 1) * 128 + 124) < 2**31      # byte offset of the last channel group of the last texel
