"""CPU-only restatements of index arithmetic that round 4's late kernels rely on (no GPU, no library): the models are the
statements of the kernels, line by line, so that a change of a constant on either side shows up here first."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize("nrows", [256, 255, 1, 37])
def test_glue_forward_staging_walk_visits_every_coefficient_once(nrows):
    """csrc/glue.hip::glue_stage_rows: thread t handles e = t + 256 i of the block's dshs slice; (q, r), the LDS word q * 49 + r
    and the f_rest index e - 3 (q + 1) advance incrementally (256 = 5 * 48 + 16, one carry).  Every element of the slice must be
    visited exactly once with exactly those values, and r >= 3 must address f_rest[45 q + r - 3]."""
    ROW = 49
    seen = np.zeros(nrows * 48, dtype=np.int32)
    for t in range(256):
        r, q = t % 48, t // 48
        word, src = q * ROW + r, t - 3 * (q + 1)
        e = t
        while e < nrows * 48:
            q, r_true = divmod(e, 48)              # the kernel carries r, word and src only
            assert r == r_true
            assert word == q * ROW + r
            if r >= 3:
                assert src == 45 * q + (r - 3) and 0 <= src < nrows * 45
            else:
                assert max(src, 0) < max(nrows * 45, 1)          # the unconditional load of the kernel stays inside f_rest
            seen[e] += 1
            r += 16; word += 5 * ROW + 16; src += 256 - 15
            if r >= 48:
                r -= 48; word += ROW - 48; src -= 3
            e += 256
    assert (seen == 1).all()
    # read-back: lane t reads words 49 t + k, k < 48 -- 64 consecutive lanes hit 64 different banks of a 64-bank LDS
    for k in (0, 17, 47):
        assert len({(ROW * t + k) % 64 for t in range(64)}) == 64


@pytest.mark.parametrize("tiles", [1, 1023, 1024, 1025, 6700, 38000, 129600])
def test_scan_tiles_partition_covers_every_tile_once(tiles):
    """csrc/raster_forward.hip::scan_tiles_kernel: thread t owns tiles [t * per, min(tiles, (t + 1) * per)), per = ceil(tiles / 1024);
    ranges = exclusive scan in tile order."""
    g = np.random.default_rng(tiles)
    cnt = g.integers(0, 500, size=tiles).astype(np.uint32)
    per = (tiles + 1023) // 1024
    mine = np.zeros(1024, dtype=np.uint64)
    owner = np.full(tiles, -1)
    for t in range(1024):
        t0, t1 = t * per, min(tiles, (t + 1) * per)
        if t0 < t1:
            assert (owner[t0:t1] == -1).all()
            owner[t0:t1] = t
            mine[t] = cnt[t0:t1].sum()
    assert (owner >= 0).all()
    start = np.concatenate(([0], np.cumsum(mine)[:-1]))
    ranges = np.zeros((tiles, 2), dtype=np.uint64)
    for t in range(1024):
        s = start[t]
        for i in range(t * per, min(tiles, (t + 1) * per)):
            ranges[i] = (s, s + cnt[i])
            s += cnt[i]
    ref = np.concatenate(([0], np.cumsum(cnt.astype(np.uint64))))
    assert (ranges[:, 0] == ref[:-1]).all() and (ranges[:, 1] == ref[1:]).all()


@pytest.mark.parametrize("P", [1, 63, 100_000, 1_200_000, 2_500_000])
def test_binning_chunks_cover_the_gaussians_in_whole_waves(P):
    """csrc/common.hpp::bin_blocks / bin_chunk (round 4: chunks rounded to 64, not to 256): contiguous, disjoint, complete."""
    MAX_BIN_BLOCKS = 512
    nb = min(max((P + 255) // 256, 1), MAX_BIN_BLOCKS)
    per = (P + nb - 1) // nb
    chunk = (per + 63) // 64 * 64
    assert chunk % 64 == 0 and nb * chunk >= P
    covered = 0
    for b in range(nb):
        g0, g1 = b * chunk, min(P, b * chunk + chunk)
        covered += max(0, g1 - g0)
    assert covered == P
    if P == 1_200_000:
        assert sum(1 for b in range(nb) if b * chunk < P) == 507        # 469 with chunks rounded to 256


def test_no_backward_predicate_of_the_autograd_nodes():
    """rasterizer._no_backward decides `forward_only` BEFORE Function.apply (inside forward() grad mode is always off)."""
    from s3gaussian_amd.rasterizer import _no_backward
    a, b = torch.zeros(3), torch.zeros(3, requires_grad=True)
    assert _no_backward(a, None, torch.Tensor([]))
    assert not _no_backward(a, b)
    with torch.no_grad():
        assert _no_backward(a, b)
    assert not _no_backward(b.detach().requires_grad_(True))

