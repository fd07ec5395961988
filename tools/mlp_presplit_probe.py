"""Round 5: the pre-split bf16x3 training kernels (S3G_MLP_BF16X3) against the on-the-fly ones they replace (S3G_MLP_BF16X3_ONTHEFLY)
-- outputs, activation stash and mask words (forward), g_features, gradient signals and all sixteen weight gradients (backward) must be
BIT-IDENTICAL: same pieces, same MFMA order -- and against the exact fp32 chain for time.      python tools/mlp_presplit_probe.py [P] [reps]
One JSON line."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from s3gaussian_amd import _lib, mlp  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_013
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
L = mlp._bind()
L.s3g_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
g = torch.Generator().manual_seed(0)
x = (torch.rand(P, 128, generator=g) * 0.5).to(dev)
params = [(torch.randn(*s, generator=g) * (0.2 if len(s) == 2 else 0.05)).to(dev) for s in mlp._SHAPES]
g_dx, g_dshs, g_feat = (torch.randn(P, n, generator=g).to(dev) for n in (3, 48, 3))
pack_floats = L.s3g_deform_mlp_pack_bytes() // 4


def run(mode):
    mlp.set_mlp_arithmetic(mode)
    dx, dshs, feat = (torch.empty(P, n, device=dev) for n in (3, 48, 3))
    stash = torch.zeros(L.s3g_deform_mlp_stash_bytes(P) // 4, device=dev)
    w = mlp._pack(params)
    grads = [torch.zeros_like(p) for p in params]
    gw = mlp._pack(grads)
    gx = torch.empty_like(x)
    ws = torch.empty(5, P, 64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    times = {}
    for r in range(reps + 1):
        if r == 1:
            for i in (5, 6, 7):
                L.s3g_profile_read(i, None, None, None)
            L.s3g_profile_enable(1)
        for gr in grads:
            gr.zero_()
        _lib.check(L.s3g_deform_mlp_forward(C.byref(w), P, x.data_ptr(), dx.data_ptr(), dshs.data_ptr(), feat.data_ptr(), stash.data_ptr(), 1, stream))
        _lib.check(L.s3g_deform_mlp_backward(C.byref(w), P, x.data_ptr(), stash.data_ptr(), g_dx.data_ptr(), g_dshs.data_ptr(), g_feat.data_ptr(),
                                             gx.data_ptr(), C.byref(gw), ws.data_ptr(), stream))
    torch.cuda.synchronize()
    L.s3g_profile_enable(0)
    for i, name in ((5, "forward_ms"), (6, "backward_ms"), (7, "wgrad_ms")):
        ms = C.c_double()
        n = L.s3g_profile_read(i, C.byref(ms), None, None)
        times[name] = round(ms.value / max(n, 1), 4)
    return dict(dx=dx, dshs=dshs, feat=feat, stash=stash[pack_floats:].clone(), gx=gx, ws=ws, grads=grads), times


def forward_only(mode, save):
    """the forward alone, with / without the activation stash (save = 0: what an inference render that draws the feature image runs)"""
    mlp.set_mlp_arithmetic(mode)
    dx, dshs, feat = (torch.empty(P, n, device=dev) for n in (3, 48, 3))
    stash = torch.zeros(L.s3g_deform_mlp_stash_bytes(P) // 4, device=dev)
    w = mlp._pack(params)
    stream = torch.cuda.current_stream().cuda_stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 2)]
    for r in range(reps + 1):
        ev[r].record()
        _lib.check(L.s3g_deform_mlp_forward(C.byref(w), P, x.data_ptr(), dx.data_ptr(), dshs.data_ptr(), feat.data_ptr(), stash.data_ptr(), save, stream))
    ev[reps + 1].record()
    torch.cuda.synchronize()
    return round(ev[1].elapsed_time(ev[reps + 1]) / reps, 4)     # (includes the pack kernels: ~10 us)


res, times = {}, {}
for mode in ("f32", "bf16x3_onthefly", "bf16x3"):
    res[mode], times[mode] = run(mode)
    times[mode]["forward_call_ms_with_stash"] = forward_only(mode, 1)
    times[mode]["forward_call_ms_no_stash"] = forward_only(mode, 0)
mlp.set_mlp_arithmetic("f32")
a, b = res["bf16x3_onthefly"], res["bf16x3"]
bits = lambda t: t.contiguous().view(torch.int32)       # (mask words reinterpreted as floats may be NaN patterns: compare the bits)
same = {k: bool(torch.equal(bits(a[k]), bits(b[k]))) for k in ("dx", "dshs", "feat", "stash", "gx", "ws")}
same["weight_grads_max_rel"] = max(float((ga - gb).abs().max() / ga.abs().max().clamp_min(1e-30)) for ga, gb in zip(a["grads"], b["grads"]))
f = res["f32"]
dist = {k: float((b[k] - f[k]).norm() / f[k].norm()) for k in ("dx", "dshs", "feat", "gx")}
print(json.dumps(dict(P=P, reps=reps, times_ms=times, presplit_equals_onthefly=same, rel_l2_bf16x3_vs_f32=dist)))
