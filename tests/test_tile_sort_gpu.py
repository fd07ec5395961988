"""The per-tile sort at the boundaries of its code paths (csrc/raster_forward.hip::sort_tiles_kernel, round 6): ONE 16x16 tile holding
exactly n instances, n on both sides of every switch -- <= 256 keys ranked by counting, <= 512 on the bitonic network, (512, 4096] bucket
pass in the medium launch, (4096, 7616] bucket pass in the long-list launch, <= 16384 network in LDS, longer lists network in global
memory -- and with depth ties of every size (the index decides; a tie of thousands of instances is ONE bucket).  The expected list is
the ascending (depth bits, Gaussian index) order of the visible Gaussians, from the kernel's own depths (reference:
rasterizer_impl.cu:duplicateWithKeys + the stable radix sort, i.e. tests/test_raster_ref_gpu.py's bar, here without the reference build
so that sizes the small scenes never reach are covered too)."""
import numpy as np
import pytest
import torch

from tests.util import tiny_scene

pytestmark = pytest.mark.gpu
SIZES = [1, 2, 255, 256, 257, 511, 512, 513, 1023, 1025, 4095, 4096, 4097, 7615, 7616, 7617, 7700, 16383, 16384, 16385, 20011]


def _point_list(s, dev):
    from diff_gaussian_rasterization import _C
    from s3gaussian_amd import _debug
    cam = s["cam"]
    e = torch.Tensor([])
    P = s["means3D"].shape[0]
    H, W = cam["image_height"], cam["image_width"]
    d = lambda k: s[k].to(dev).contiguous()
    R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(
        s["bg"].to(dev), d("means3D"), d("colors_precomp"), d("opacities"), d("scales"), d("rotations"), 1.0, e,
        cam["viewmatrix"].to(dev), cam["projmatrix"].to(dev), cam["tanfovx"], cam["tanfovy"], H, W, e, 0, cam["campos"].to(dev), False, False)
    g, im, b = _debug.decode_geometry(geom, P), _debug.decode_image(img, W, H), _debug.decode_binning(binning, R)
    return R, radii.cpu().numpy(), g["depths"].cpu().numpy(), im["ranges"].cpu().numpy(), b["point_list"].cpu().numpy().astype(np.int64)


TIE_SIZES = [256, 512, 1025, 4096, 7616, 7700, 16384, 20011]       # one size per code path for the tie patterns
CASES = [(n, "none") for n in SIZES] + [(n, t) for n in TIE_SIZES for t in ("pairs", "two_depths", "one_depth")]


@pytest.mark.parametrize("n,ties", CASES)
def test_one_tile_of_exactly_n_instances_is_sorted_by_depth_then_index(gpu_device, n, ties):
    s = tiny_scene(P=n, W=16, H=16, seed=n % 97 + 3, scale=0.004, sigma=0.1, spread=0.05)     # all in the middle of the one tile
    s["opacities"] = s["opacities"] * 0.01 + 0.005
    z = s["means3D"][:, 2]
    if ties == "pairs":
        z[1::2] = z[0::2][: z[1::2].shape[0]]
    elif ties == "two_depths":
        z[:] = 4.0
        z[::3] = 4.5
    elif ties == "one_depth":
        z[:] = 5.0
    R, radii, depths, ranges, pl = _point_list(s, gpu_device)
    vis = np.nonzero(radii > 0)[0]
    assert R == vis.size == n and ranges.shape[0] == 1 and tuple(ranges[0]) == (0, n)
    bits = depths.view(np.uint32).astype(np.uint64)[vis]
    expected = vis[np.argsort((bits << np.uint64(32)) | vis.astype(np.uint64), kind="stable")]
    np.testing.assert_array_equal(pl[:n], expected)
