"""The lane algebra of raster_backward.hip::row_transpose_reduce, simulated: 13 (or 10) values per lane, 16 lanes per row,
29 masked DPP adds -> every lane of bank b ends with the row totals of values 4b .. 4b+3.

DPP semantics used (gfx9): `v_add_f32_dpp dst, src, src  <perm> bank_mask:M` computes, for every lane whose bank (lane // 4 inside
its row of 16) is enabled in M, dst[lane] = src[perm(lane)] + src[lane]; other lanes keep dst.  row_ror:8 reads lane ^ 8,
row_half_mirror reads 7 - lane inside each half row, quad_perm:[1,0,3,2] reads lane ^ 1, quad_perm:[2,3,0,1] reads lane ^ 2.
Test infrastructure only (documents why the kernel's sequence is a correct reduction)."""
import numpy as np
import pytest


def dpp_add(dst, src, perm, bank_mask):
    out = dst.copy()
    for lane in range(16):
        if bank_mask >> (lane // 4) & 1:
            out[lane] = src[perm(lane)] + src[lane]
    return out


ROR8 = lambda l: l ^ 8
HALF_MIRROR = lambda l: (l & 8) | (7 - (l & 7))
QUAD_1032 = lambda l: l ^ 1
QUAD_2301 = lambda l: l ^ 2


def transpose_reduce(v, nv):
    """v: [nv][16] float64 lane values.  Mirrors the kernel statement by statement; registers the kernel leaves unwritten hold
    NaN here, so any use of them would show."""
    r = [np.full(16, np.nan) for _ in range(8)]
    for i in range(8):
        r[i] = dpp_add(r[i], v[i], ROR8, 0x3)
        if i + 8 < nv:
            r[i] = dpp_add(r[i], v[i + 8], ROR8, 0xC)
    q = [np.full(16, np.nan) for _ in range(4)]
    for i in range(4):
        q[i] = dpp_add(q[i], r[i], HALF_MIRROR, 0x5)
        q[i] = dpp_add(q[i], r[i + 4], HALF_MIRROR, 0xA)
    for perm in (QUAD_1032, QUAD_2301):
        for i in range(4):
            q[i] = dpp_add(q[i], q[i], perm, 0xF)
    return q


@pytest.mark.parametrize("nv", [10, 13])
def test_every_bank_ends_with_the_totals_of_its_four_values(nv):
    rng = np.random.default_rng(nv)
    v = rng.integers(-1000, 1000, size=(nv, 16)).astype(np.float64)     # integers: sums are exact in any order
    q = transpose_reduce(v, nv)
    for bank in range(4):
        for k in range(4):
            value = 4 * bank + k
            if value >= nv:
                continue                       # rows the kernel stores but nobody reads
            for lane in range(4 * bank, 4 * bank + 4):
                assert q[k][lane] == v[value].sum(), (bank, k, lane)


def test_add_count_is_29_for_13_values():
    assert 8 + (13 - 8) + 8 + 4 + 4 == 29      # step A (8 + 5), B (8), C (4), D (4); a plain reduction needs 4 * 13 = 52
