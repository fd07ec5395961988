"""Checkpoint / point-cloud I/O compatibility with the reference's formats (scene/gaussian_model.py:71-111 capture /
restore, :258-275 save_ply, :355-395 load_ply).  CPU only."""

import numpy as np
import torch

from s3gaussian_amd import plyio
from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt


def _small_hyper():
    return default_hyper(kplanes_config=dict(grid_dimensions=2, input_coordinate_dim=4, output_coordinate_dim=32,
                                             resolution=[4, 4, 4, 3]), multires=[1, 2])


def _pc(P=37, seed=0):
    g = torch.Generator().manual_seed(seed)
    pc = GaussianParams(3, _small_hyper())
    pc.init_from_tensors(torch.randn(P, 3, generator=g), torch.randn(P, 3, generator=g), torch.randn(P, 4, generator=g),
                         torch.randn(P, 1, generator=g), torch.randn(P, 16, 3, generator=g), "cpu")
    return pc


def test_ply_has_the_layout_plyfile_writes_and_round_trips(tmp_path):
    pc = _pc()
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    pc.save_ply(path)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] + \
            ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    assert lines[3:] == [f"property float {n}" for n in names]
    tab = np.frombuffer(body, "<f4").reshape(37, len(names))
    # SH coefficients are stored channel-major: f_rest_k = features_rest.transpose(1, 2).flatten()[k]
    np.testing.assert_array_equal(tab[:, 9:54], pc._features_rest.detach().transpose(1, 2).flatten(start_dim=1).numpy())
    np.testing.assert_array_equal(tab[:, 3:6], 0.0)
    other = GaussianParams(3, _small_hyper())
    other.load_ply(path, device="cpu")
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        a, b = getattr(pc, k), getattr(other, k)
        assert isinstance(b, torch.nn.Parameter) and a.shape == b.shape and torch.equal(a, b), k
    assert other.active_sh_degree == 3 and other.max_radii2D.shape == (37,)


def test_ply_reader_takes_ascii_big_endian_and_shuffled_property_order(tmp_path):
    names = ["rot_3", "x", "opacity", "z", "y"]
    tab = np.arange(15, dtype=np.float64).reshape(3, 5) / 7
    p1 = str(tmp_path / "a.ply")
    with open(p1, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 3\n" +
                "".join(f"property double {n}\n" for n in names) + "end_header\n")
        for r in tab:
            f.write(" ".join(repr(float(x)) for x in r) + "\n")
    n1, v1 = plyio.read_vertices(p1)
    assert n1 == names and all(np.allclose(v1[n], tab[:, k]) for k, n in enumerate(names))
    p2 = str(tmp_path / "b.ply")
    with open(p2, "wb") as f:
        f.write(("ply\nformat binary_big_endian 1.0\nelement vertex 3\n" + "".join(f"property float {n}\n" for n in names) +
                 "element face 0\nproperty list uchar int vertex_indices\nend_header\n").encode())
        f.write(tab.astype(">f4").tobytes())
    n2, v2 = plyio.read_vertices(p2)
    assert n2 == names and all(np.array_equal(v2[n], tab[:, k].astype(np.float32)) for k, n in enumerate(names))


def test_capture_restore_round_trip_through_torch_save(tmp_path):
    opt = default_opt()
    pc = _pc(P=21, seed=3)
    pc.training_setup(opt)
    for step in range(2):       # give the optimizer real state
        loss = sum((p ** 2).sum() for p in (pc._xyz, pc._features_dc, pc._opacity, pc._scaling, pc._rotation, pc._features_rest))
        loss = loss + sum((p ** 2).sum() for p in pc._deformation.parameters())
        loss.backward()
        pc.optimizer.step()
        pc.optimizer.zero_grad(set_to_none=True)
    pc.xyz_gradient_accum += 0.25
    pc.denom += 2
    pc.max_radii2D += 3
    tup = pc.capture()
    assert len(tup) == 14 and isinstance(tup[2], dict) and tup[0] == 3          # the reference's tuple shape
    path = str(tmp_path / "chkpnt_fine_30000.pth")
    torch.save((tup, 30000), path)
    model_args, it = torch.load(path, weights_only=False)
    back = GaussianParams(3, _small_hyper())
    back.restore(model_args, opt)
    assert it == 30000
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "max_radii2D", "xyz_gradient_accum",
              "denom", "_deformation_table"):
        assert torch.equal(getattr(pc, k), getattr(back, k)), k
    for (ka, a), (kb, b) in zip(pc._deformation.state_dict().items(), back._deformation.state_dict().items()):
        assert ka == kb and torch.equal(a, b)
    sa, sb = pc.optimizer.state_dict(), back.optimizer.state_dict()
    assert [g["name"] for g in sa["param_groups"]] == [g["name"] for g in sb["param_groups"]]
    for k in sa["state"]:
        for f in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(torch.as_tensor(sa["state"][k][f]), torch.as_tensor(sb["state"][k][f])), (k, f)
    # the restored model keeps training: one more identical step on both gives identical parameters
    for m in (pc, back):
        (m._xyz ** 2).sum().backward()
        m.optimizer.step()
    assert torch.equal(pc._xyz, back._xyz)
