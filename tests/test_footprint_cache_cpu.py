"""The walker's remembered footprint of hexplane.hip::foot1_add_t (hit path, evict, shift down / right), modelled statement by
statement in Python: whatever sequence of bilinear taps a walker sees, the flushed sums plus the final flush must equal the
direct scatter.  Integer-valued gradients and weights make every order of summation exact.  Test infrastructure only
(documents why the miss path conserves every contribution; the compiled kernel is checked by tests/test_hexplane_gpu.py).
(Rounds 1-4 carried a two-entry cache with an MRU bit, foot2_add; it went with the finest-level walk orders in round 5.  The model
below takes the gradient already divided by the sample: the division is a per-call scalar and does not touch the bookkeeping.)"""
import numpy as np
import pytest


class Foot1:
    def __init__(self, W, H, row):
        self.W, self.H, self.row = W, H, row
        self.key = -1                  # (texel index << 2 | corner flags), -1 = empty
        self.a = [0.0] * 4
        self.out = np.zeros(W * H)
        self.atomics = 0

    def _atomic(self, texel, v):
        self.out[texel] += v
        self.atomics += 1

    def add(self, key, flags, w, g):
        """foot1_add_t<ROW>: key = nw texel, flags bit0 = ne column in range, bit1 = sw row in range, w = (w00, w01, w10, w11)."""
        W, row = self.W, self.row
        tkf = (key << 2) | (flags & 3)
        if tkf != self.key:
            KF = self.key
            K, FL = KF >> 2, KF & 3
            down = (not row) and KF >= 0 and key == K + W
            right = KF >= 0 and key == K + 1 and bool(FL & 1)
            shift = down or right
            A = list(self.a)
            if row:
                A[2] = A[3] = 0.0
            if KF >= 0:
                self._atomic(K, A[0])
                if (FL & 1) and not right:
                    self._atomic(K + 1, A[1])
                if (not row) and (FL & 2) and not down:
                    self._atomic(K + W, A[2])
                if (not row) and (FL & 3) == 3 and not shift:
                    self._atomic(K + W + 1, A[3])
            self.a[0] = A[2] if down else (A[1] if right else 0.0)
            self.a[1] = A[3] if down else 0.0
            if not row:
                self.a[2] = A[3] if right else 0.0
                self.a[3] = 0.0
            self.key = tkf
        for k in range(2 if row else 4):
            self.a[k] += g * w[k]

    def flush_all(self):
        if self.key < 0:
            return
        K, FL, A = self.key >> 2, (self.key & 1) if self.row else (self.key & 3), self.a
        self._atomic(K, A[0])
        if FL & 1:
            self._atomic(K + 1, A[1])
        if FL & 2:
            self._atomic(K + self.W, A[2])
        if (FL & 3) == 3:
            self._atomic(K + self.W + 1, A[3])


def walk(rng, W, H, n, mode):
    """(x0, y0) of n consecutive taps: a walk along rows ("down"), along columns ("right"), a random jumpy one, one that
    alternates between two footprints, and one hugging the last column / row (flags clear)."""
    if mode == "down":
        x0 = np.full(n, rng.integers(0, W - 1)); y0 = np.minimum(np.arange(n) // 3, H - 1)
    elif mode == "right":
        y0 = np.full(n, rng.integers(0, max(H - 1, 1))); x0 = np.minimum(np.arange(n) // 2, W - 1)
    elif mode == "alternate":
        x0 = np.full(n, 2); y0 = 3 + (np.arange(n) % 2)
    elif mode == "border":
        x0 = np.where(np.arange(n) % 2 == 0, W - 1, W - 2); y0 = np.minimum(np.arange(n) // 2, H - 1)
    else:
        x0 = rng.integers(0, W, n); y0 = rng.integers(0, H, n)
    return x0, np.minimum(y0, H - 1)


@pytest.mark.parametrize("mode", ["down", "right", "alternate", "border", "random"])
@pytest.mark.parametrize("row", [False, True])
def test_cache_conserves_every_contribution(mode, row):
    rng = np.random.default_rng(hash((mode, row)) % 2**32)
    W, H = 9, (1 if row else 7)
    n = 40
    x0, y0 = walk(rng, W, H, n, mode)
    cache = Foot1(W, H, row)
    want = np.zeros(W * H)
    for i in range(n):
        key = int(y0[i]) * W + int(x0[i])
        flags = (1 if x0[i] + 1 < W else 0) | (2 if y0[i] + 1 < H else 0)
        w = rng.integers(1, 5, 4).astype(float)
        if not flags & 1:
            w[1] = w[3] = 0.0        # an out-of-range corner has weight exactly 0 (make_tap)
        if not flags & 2:
            w[2] = w[3] = 0.0
        g = float(rng.integers(-9, 10))
        cache.add(key, flags, w, g)
        want[key] += g * w[0]
        if flags & 1:
            want[key + 1] += g * w[1]
        if flags & 2:
            want[key + W] += g * w[2]
        if flags == 3:
            want[key + W + 1] += g * w[3]
    cache.flush_all()
    assert np.array_equal(cache.out, want)


def test_shift_halves_the_atomics_of_a_walk_down_a_column():
    W, H, n = 9, 40, 120
    cache = Foot1(W, H, False)
    for i in range(n):
        y = min(i // 3, H - 2)
        cache.add(y * W + 4, 3, (1.0, 1.0, 1.0, 1.0), 1.0)
    cache.flush_all()
    steps = min((n - 1) // 3, H - 2)
    assert cache.atomics == 2 * steps + 4          # two per step instead of four, plus the final footprint


def test_alternating_footprints_cost_an_eviction_each():
    """What the second entry of rounds 1-4 was for: a walk in a FOREIGN level's order alternates between two footprints and a
    single entry then flushes on every step.  Per-level walk orders (one order per orientation and level) never do that, which is
    why one entry suffices -- this pins the cost the sort orders must keep avoiding."""
    W, H, n = 9, 8, 40
    cache = Foot1(W, H, False)
    for i in range(n):
        cache.add((3 + 2 * (i % 2)) * W + 2, 3, (1.0, 1.0, 1.0, 1.0), 1.0)      # two footprints two rows apart: no shift reuse
    cache.flush_all()
    assert cache.atomics == 4 * n
