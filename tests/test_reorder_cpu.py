"""GaussianParams.reorder_spatially(): every per-Gaussian tensor, the Adam moments and the densification accumulators move
together, the Parameter objects survive, the Morton keys come out sorted.  Runs on CPU tensors (torch.optim.Adam)."""
import torch

from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt


def test_reorder_moves_everything_together():
    torch.manual_seed(0)
    P = 500
    pc = GaussianParams(3, default_hyper())
    xyz = torch.rand(P, 3) * 4 - 2
    tag = torch.arange(P, dtype=torch.float32)
    shs = torch.zeros(P, 16, 3)
    shs[:, 0, 0] = tag
    pc.init_from_tensors(xyz, tag[:, None].repeat(1, 3).clone(), torch.rand(P, 4), tag[:, None].clone(), shs, torch.device("cpu"))
    pc.training_setup(default_opt())
    for p in (pc._xyz, pc._opacity, pc._features_dc):                  # some optimizer state that is a function of the tag
        p.grad = torch.ones_like(p) * tag.view(-1, *([1] * (p.dim() - 1)))
    for g in pc.optimizer.param_groups:
        g["lr"] = 0.0                                                  # moments are built, parameters stay what they are
    pc.optimizer.step()
    pc.max_radii2D = tag.clone()
    pc.xyz_gradient_accum = tag[:, None].clone()
    ids = {id(p) for p in (pc._xyz, pc._opacity, pc._scaling, pc._rotation, pc._features_dc, pc._features_rest)}
    old_xyz = pc._xyz.detach().clone()
    m_before = pc.optimizer.state[pc._opacity]["exp_avg"].clone()
    pc._scaling.grad = tag[:, None].repeat(1, 3).clone()              # a gradient still pending when the reorder happens
    perm = pc.reorder_spatially()
    assert torch.equal(pc._scaling.grad[:, 0], tag[perm])              # ... follows its Gaussian
    assert sorted(perm.tolist()) == list(range(P))
    assert {id(p) for p in (pc._xyz, pc._opacity, pc._scaling, pc._rotation, pc._features_dc, pc._features_rest)} == ids
    new_tag = pc._opacity.detach()[:, 0]
    assert torch.equal(new_tag, tag[perm])                             # opacity carries the old index
    assert torch.equal(pc._xyz.detach(), old_xyz[perm])
    assert torch.equal(pc._scaling.detach()[:, 0], new_tag) and torch.equal(pc._features_dc.detach()[:, 0, 0], new_tag)
    assert torch.equal(pc.max_radii2D, new_tag) and torch.equal(pc.xyz_gradient_accum[:, 0], new_tag)
    assert torch.equal(pc.optimizer.state[pc._opacity]["exp_avg"], m_before[perm])
    assert pc._deformation_table.shape[0] == P
    # Morton order: the 3-D cell keys are non-decreasing, and a second call is the identity permutation
    assert torch.equal(pc.reorder_spatially(), torch.arange(P))
