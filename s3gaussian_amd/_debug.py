"""Decode the private scratch arenas of libs3g.so into named tensors (tests / profiling only).

Mirrors the carving order of s3gaussian_amd/csrc/common.hpp (GeomState / ImageState / BinningState, 128-byte
aligned sub-arrays).  The layout is private to a build: never rely on it in product code.
"""
from __future__ import annotations

import torch


class _Carver:
    def __init__(self, buf: torch.Tensor):
        self.buf, self.off = buf, 0

    def take(self, count: int, dtype: torch.dtype):
        self.off = (self.off + 127) & ~127
        nbytes = count * torch.empty(0, dtype=dtype).element_size()
        out = self.buf[self.off:self.off + nbytes].view(dtype)
        self.off += nbytes
        return out


def decode_geometry(buf: torch.Tensor, P: int) -> dict:
    c = _Carver(buf)
    return dict(depths=c.take(P, torch.float32), means2D=c.take(2 * P, torch.float32).view(P, 2),
                conic_opacity=c.take(4 * P, torch.float32).view(P, 4), cov3D=c.take(6 * P, torch.float32).view(P, 6),
                rgb=c.take(3 * P, torch.float32).view(P, 3), clamped=c.take(3 * P, torch.uint8).view(P, 3),
                rect=c.take(4 * P, torch.int16).view(P, 4), gauss_off=c.take(P, torch.int32), tile_mask=c.take(P, torch.int32))


def decode_image(buf: torch.Tensor, W: int, H: int) -> dict:
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    c = _Carver(buf)
    return dict(final_T=c.take(W * H, torch.float32).view(H, W), n_contrib=c.take(W * H, torch.int32).view(H, W),
                ranges=c.take(2 * tiles, torch.int32).view(tiles, 2), tile_count=c.take(tiles, torch.int32),
                tile_hi=c.take(tiles, torch.int32), ctrl=c.take(8, torch.int32))


def decode_binning(buf: torch.Tensor, R: int) -> dict:
    """slot_pos is the last array: one entry per tile of every Gaussian's rect (S >= R entries, -1 = culled tile)."""
    c = _Carver(buf)
    keys, point_list = c.take(R, torch.int64), c.take(R, torch.int32)
    c.off = (c.off + 127) & ~127
    slot_pos = buf[c.off:c.off + ((buf.numel() - c.off) // 4) * 4].view(torch.int32)
    return dict(keys=keys, point_list=point_list, slot_pos=slot_pos)
