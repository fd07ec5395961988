"""Line-atomics of the HexPlane scatter under alternative walk orders, on the bench's own point cloud (CPU only).
    python tools/sim/flush_orders.py [P]
Orders: row-major finest-level cells (the kernel's today), one row-major order PER LEVEL, Morton order of the finest cells."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from s3gaussian_amd import synth  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
SIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flush_sim")
if not os.path.exists(SIM) or os.path.getmtime(SIM) < os.path.getmtime(SIM + ".c"):
    subprocess.check_call(["gcc", "-O2", "-o", SIM, SIM + ".c"])
sc = synth.street_scene(P=P, seed=0, width=1600, height=1066, n_frames=2)
xyz = sc["gaussians"]["xyz"].numpy().astype(np.float32)
amax, amin = (np.asarray(v, np.float32) for v in sc["aabb"])
u = (xyz - amax) * (np.float32(2.0) / (amin - amax)) - np.float32(1.0)       # point_coords()
RES = [64, 128, 256, 512]
MAJ, MIN_ = [0, 1, 2], [1, 2, 0]
# plane (axw, axh) of orientation o: PLA = (x,y) (y,z) (x,z) -> PAIR0/PAIR1 of plane ids 0, 3, 1
PLANE_AXES = {0: (0, 1), 1: (1, 2), 2: (0, 2)}


def cell(axis, W):
    ix = (u[:, axis] + 1) / 2 * np.float32(W - 1)
    ix = np.clip(ix, 0, W - 1)
    return np.floor(ix).astype(np.int64)


def run(keys, flags, W, row, seg, entries):
    with tempfile.TemporaryDirectory() as d:
        kf, ff = os.path.join(d, "k"), os.path.join(d, "f")
        keys.astype(np.int32).tofile(kf)
        flags.astype(np.uint8).tofile(ff)
        out = subprocess.check_output([SIM, kf, ff, str(len(keys)), str(W), str(int(row)), str(seg), str(entries)]).split()
    return [int(x) for x in out]


def footprint(o, level, order):
    axw, axh = PLANE_AXES[o]
    W = RES[level]
    x0, y0 = cell(axw, W)[order], cell(axh, W)[order]
    return y0 * W + x0, ((x0 + 1 < W) * 1 + (y0 + 1 < W) * 2), W


def rowfoot(o, level, order):
    W = RES[level]
    x0 = cell(MAJ[o], W)[order]
    return x0, (x0 + 1 < W) * 1, W


def morton(a, b):
    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        return (v | (v << 1)) & 0x55555555
    return spread(a) | (spread(b) << 1)


def report(name, total, parts):
    print(f"{name:58s} {total / 1e6:7.2f} M line-atomics = {total / P:5.2f} per point   " + "  ".join(f"{k} {v / 1e6:.2f}" for k, v in parts.items()), flush=True)


for seg in (256, 1024):
    # (a) today: one order per orientation (major, minor finest cell); spatial + time-row plane of every level in it; two entries
    tot, parts = 0, {"spatial": 0, "rows": 0, "seg_end": 0}
    for o in range(3):
        order = np.lexsort((cell(MIN_[o], 512), cell(MAJ[o], 512)))
        for l in range(4):
            k, f, W = footprint(o, l, order)
            a_, _, _, se = run(k, f, W, False, seg, 2)
            parts["spatial"] += a_; parts["seg_end"] += se
            k, f, W = rowfoot(o, l, order)
            a_, _, _, se = run(k, f, W, True, seg, 2)
            parts["rows"] += a_; parts["seg_end"] += se
    report(f"today: finest row-major order, 2 entries, seg {seg}", parts["spatial"] + parts["rows"], parts)
    for entries in (1, 2):
        # (b) one row-major order per (orientation, level): that level's own cells
        tot, parts = 0, {"spatial": 0, "rows": 0, "seg_end": 0}
        for o in range(3):
            for l in range(4):
                W = RES[l]
                order = np.lexsort((cell(MIN_[o], W), cell(MAJ[o], W)))
                k, f, W = footprint(o, l, order)
                a_, _, _, se = run(k, f, W, False, seg, entries)
                parts["spatial"] += a_; parts["seg_end"] += se
                k, f, W = rowfoot(o, l, order)
                a_, _, _, se = run(k, f, W, True, seg, entries)
                parts["rows"] += a_; parts["seg_end"] += se
        report(f"per-level row-major orders, {entries} entr{'y' if entries == 1 else 'ies'}, seg {seg}", parts["spatial"] + parts["rows"], parts)
    # (c) Morton order of the finest cells for the spatial planes; rows in the major-sorted order of today
    parts = {"spatial": 0, "rows": 0, "seg_end": 0}
    for o in range(3):
        axw, axh = PLANE_AXES[o]
        order = np.argsort(morton(cell(axw, 512), cell(axh, 512)), kind="stable")
        for l in range(4):
            k, f, W = footprint(o, l, order)
            a_, _, _, se = run(k, f, W, False, seg, 2)
            parts["spatial"] += a_; parts["seg_end"] += se
        order = np.lexsort((cell(MIN_[o], 512), cell(MAJ[o], 512)))
        for l in range(4):
            k, f, W = rowfoot(o, l, order)
            a_, _, _, se = run(k, f, W, True, seg, 2)
            parts["rows"] += a_; parts["seg_end"] += se
    report(f"Morton (finest cells) for the spatial planes, seg {seg}", parts["spatial"] + parts["rows"], parts)
# floor: distinct footprints
dist = 0
for o in range(3):
    for l in range(4):
        k, f, W = footprint(o, l, np.arange(P))
        dist += len(np.unique(k))
print(f"distinct spatial footprints over the 12 plane-levels: {dist / 1e6:.2f} M (x 2..4 atomics each = the floor of any walk)")
