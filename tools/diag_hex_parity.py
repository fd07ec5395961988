import sys, os, torch
sys.path.insert(0, os.getcwd())
import tests.test_parity_fullsize_gpu as T
from s3gaussian_amd import synth
dev = torch.device("cuda:0")
sc = synth.street_scene(P=1_200_000, seed=0, n_frames=4)
street = dict(xyz=sc["gaussians"]["xyz"].to(dev), aabb=sc["aabb"])
P = 70000
ref, mine = T._fields(dev, street["aabb"], seed=P % 1000)
g = torch.Generator().manual_seed(P + 7)
xyz = street["xyz"][:P].clone()
moved = torch.zeros(P, dtype=torch.bool, device=dev); moved[::97] = True
xyz[::97] += torch.tensor([150.0, -60.0, 20.0], device=dev) * (torch.rand(xyz[::97].shape[0], 1, generator=g).to(dev) - 0.5)
for tmode in ("uniform", "per_point"):
    time = (torch.rand(P, 1, generator=g) * 1.2 - 0.1).to(dev) if tmode == "per_point" else torch.full((P, 1), 0.37, device=dev)
    fr = ref(xyz.double(), time.double())
    fg = mine(xyz, time, uniform_time=(tmode == "uniform"))
    err = (fg.double() - fr).abs()
    print(tmode, "max abs err all", float(err.max()), "moved pts", float(err[moved].max()), "unmoved", float(err[~moved].max()))
    i = int(err.argmax()); p, c = divmod(i, 128)
    print("  worst at point", p, "chan", c, "ref", float(fr[p, c]), "got", float(fg[p, c]), "xyz", xyz[p].tolist(), "t", float(time[p]), "moved", bool(moved[p]))
    rel = ((err - 1e-6).clamp_min(0) / fr.abs().clamp_min(1e-300))
    i = int(rel.argmax()); p, c = divmod(i, 128)
    print("  worst rel at point", p, "chan", c, "ref", float(fr[p, c]), "got", float(fg[p, c]), "xyz", xyz[p].tolist(), "t", float(time[p]), "moved", bool(moved[p]))
    print("  ref abs max", float(fr.abs().max()), "mean", float(fr.abs().mean()))
    # fp32 torch on the GPU for comparison
    ref32 = T._fields(dev, street["aabb"], seed=P % 1000)[0].float()
    f32 = ref32(xyz, time)
    e32 = (f32.double() - fr).abs()
    print("  torch fp32 GPU grid_sample vs fp64: max abs err", float(e32.max()), " product vs torch fp32:", float((fg - f32).abs().max()))
