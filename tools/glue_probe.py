"""Render glue (activations + SH -> RGB) at cfg3 size: forward with / without the |dshs| regulariser sum, and backward, timed with
stream events; prints one JSON line.  `S3G_LIB_PATH` selects a variant build (tools/mkvariants.py).  Also checks that the colours of
two launches are bit-identical and reports a checksum, so that two builds can be compared across calls."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from s3gaussian_amd.glue import activations_and_colors
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1200000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)
    f_dc, f_rest, dshs, xyz = mk(P, 1, 3), 0.3 * mk(P, 15, 3), 0.1 * mk(P, 16, 3), 3 * mk(P, 3)
    ls, rr, ol, campos = 0.5 * mk(P, 3), mk(P, 4), mk(P, 1), torch.tensor([0.3, -0.2, 1.1], device=dev)
    out = {"P": P, "lib": os.environ.get("S3G_LIB_PATH", "tree")}

    def timed(fn):
        for _ in range(3):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        torch.cuda.synchronize()
        ev[0].record()
        for i in range(reps):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
        return round(1000.0 * t[len(t) // 2], 1)

    with torch.no_grad():
        out["forward_us"] = timed(lambda: activations_and_colors(3, f_dc, f_rest, dshs, xyz, campos, ls, rr, ol))
        out["forward_no_dshs_us"] = timed(lambda: activations_and_colors(3, f_dc, f_rest, None, xyz, campos, ls, rr, ol))
        c1 = activations_and_colors(3, f_dc, f_rest, dshs, xyz, campos, ls, rr, ol)[0]
        c2 = activations_and_colors(3, f_dc, f_rest, dshs, xyz, campos, ls, rr, ol)[0]
        out["bit_identical_two_launches"] = bool(torch.equal(c1, c2))
        out["colors_checksum"] = float(c1.double().sum())
        out["colors_xor"] = int(c1.view(torch.int32).to(torch.int64).sum())
    leaves = [t.clone().requires_grad_(True) for t in (f_dc, f_rest, dshs, xyz, ls, rr, ol)]

    def fwd_l1():
        with torch.no_grad():
            return activations_and_colors(3, f_dc, f_rest, dshs, xyz, campos, ls, rr, ol, with_dshs_l1=True)
    out["forward_with_l1_us"] = timed(fwd_l1)   # no_grad: the kernel still sums |dshs| when asked to
    o = activations_and_colors(3, *leaves[:3], leaves[3], campos, *leaves[4:], with_dshs_l1=True)
    out["dshs_l1"] = float(o[4])
    out["dshs_l1_torch"] = float(dshs.abs().mean())
    w = [torch.randn_like(x) for x in o[:4]]

    def step():
        for t in leaves:
            t.grad = None
        o = activations_and_colors(3, *leaves[:3], leaves[3], campos, *leaves[4:], with_dshs_l1=True)
        (sum((a * b).sum() for a, b in zip(o[:4], w)) + 0.5 * o[4]).backward()
    out["forward_backward_torch_sum_us"] = timed(step)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
