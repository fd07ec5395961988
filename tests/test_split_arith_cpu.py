"""The arithmetic behind S3G_MLP_BF16X3 / s3g_deform_infer_split (s3gaussian_amd/csrc/mlp.hip: split_pair, mfma_split), restated in
numpy: every fp32 number is the EXACT sum of three round-to-nearest bf16 pieces, and the six piece products of weight >= 2^-16
reproduce a product to <= 2^-23 relative.  Also pins split_feature(): the element order shared by the A fragments of the packed image
and the accumulator registers that become the B operand.  (The kernels themselves are checked on the GPU: tests/test_mlp_gpu.py,
tests/test_infer_gpu.py.)"""
import numpy as np


def rne_bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    p0 = rne_bf16(x)
    r1 = (x - p0).astype(np.float32)
    p1 = rne_bf16(r1)
    p2 = rne_bf16((r1 - p1).astype(np.float32))
    return p0, p1, p2


def test_three_bf16_pieces_are_an_exact_decomposition():
    rng = np.random.default_rng(0)
    # exact wherever the residuals stay normal numbers and the leading piece does not round up to infinity:
    # 2^-110 <= |x| < 3.39e38 (and 0); network weights, activations and gradients live far inside
    x = np.concatenate([rng.standard_normal(200_000) * 10.0 ** rng.uniform(-20, 20, 200_000), [0.0, -0.0, 1.0, -1.0, 3.3e38, 1e-30,
                        np.float32(1) + np.float32(2 ** -23), np.float32(2 ** -110)]]).astype(np.float32)
    p0, p1, p2 = split3(x)
    assert np.array_equal(p0.astype(np.float64) + p1.astype(np.float64) + p2.astype(np.float64), x.astype(np.float64))
    for p in (p0, p1, p2):                                           # each piece is a bf16 number: low 16 bits clear
        assert not (p.view(np.uint32) & 0xFFFF).any()
    nz = x != 0
    assert (np.abs(p1[nz]) <= np.abs(x[nz]) * 2.0 ** -8).all() and (np.abs(p2[nz]) <= np.abs(x[nz]) * 2.0 ** -16).all()
    # below that range the decomposition degrades gracefully (absolute error under the smallest normal number) ...
    tiny = np.array([2.0 ** -126, 3e-39, 1.1e-36], np.float32)
    q = split3(tiny)
    assert (np.abs(q[0].astype(np.float64) + q[1] + q[2] - tiny.astype(np.float64)) <= 2.0 ** -133).all()
    # ... and above it the leading piece is infinite, as the bf16 conversion of any such number is
    with np.errstate(invalid="ignore", over="ignore"):
        assert np.isinf(split3(np.array([3.4e38], np.float32))[0]).all()


def test_six_piece_products_carry_a_product_to_2_to_the_minus_23():
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(100_000) * 10.0 ** rng.uniform(-3, 3, 100_000)).astype(np.float32)
    b = (rng.standard_normal(100_000) * 10.0 ** rng.uniform(-3, 3, 100_000)).astype(np.float32)
    A, B = [p.astype(np.float64) for p in split3(a)], [p.astype(np.float64) for p in split3(b)]
    kept = sum(A[i] * B[j] for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)))       # mfma_split's six terms
    exact = a.astype(np.float64) * b.astype(np.float64)
    assert (np.abs(kept - exact) <= 2.0 ** -23 * np.abs(exact)).all()
    for i, j in ((0, 0), (0, 1), (1, 1), (2, 0)):                    # every piece product is exact in fp32 (8 x 8 significand bits)
        assert np.array_equal((A[i] * B[j]).astype(np.float32).astype(np.float64), A[i] * B[j])


def test_dot_products_have_fp32_accuracy():
    rng = np.random.default_rng(2)
    W = rng.standard_normal((64, 128)).astype(np.float32) * 0.2
    x = rng.standard_normal((128, 512)).astype(np.float32)
    Ws, xs = [p.astype(np.float64) for p in split3(W)], [p.astype(np.float64) for p in split3(x)]
    split = sum(Ws[i] @ xs[j] for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)))
    ref = W.astype(np.float64) @ x.astype(np.float64)
    fp32 = W @ x                                                     # an fp32 chain, for scale
    err_split = np.abs(split - ref).max() / np.abs(ref).max()
    err_fp32 = np.abs(fp32.astype(np.float64) - ref).max() / np.abs(ref).max()
    assert err_split < 2.0 ** -22 and err_split < err_fp32           # what is dropped is below an fp32 chain's own rounding


def test_split_feature_is_the_accumulator_register_order():
    """mlp.hip: register r of lane l holds feature rrow(r) + 4*(l >> 5) of point l & 31 (rrow(r) = (r & 3) + 8 * (r >> 2)); at K step
    ks = 2*mbi + s the lane's eight B elements are registers 8s .. 8s+7 of block mbi: split_feature(ks, h, e) must name them."""
    rrow = lambda r: (r & 3) + 8 * (r >> 2)
    split_feature = lambda ks, h, e: 16 * ks + 4 * h + (e & 3) + 8 * (e >> 2)
    seen = set()
    for mbi in range(4):
        for s in range(2):
            for h in range(2):
                for e in range(8):
                    f = split_feature(2 * mbi + s, h, e)
                    assert f == 32 * mbi + rrow(8 * s + e) + 4 * h
                    seen.add(f)
    assert seen == set(range(128))                                   # every input feature exactly once per (mbo) row block
