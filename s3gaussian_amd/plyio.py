"""Minimal PLY vertex-table reader / writer for Gaussian checkpoints (numpy only).

The reference saves / loads point clouds through the `plyfile` package (scene/gaussian_model.py:258-275 `save_ply`,
:355-395 `load_ply`), which is not installed here.  This module writes the same file -- one `vertex` element whose
properties are all `float` (x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_*), `binary_little_endian 1.0` like
plyfile's default -- and reads binary-little-endian / ascii vertex tables with scalar properties, so files interchange
with the reference's in both directions.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}


def write_vertices(path: str, names: List[str], table: np.ndarray) -> None:
    """table [N, len(names)] -> float32 properties, binary little endian (what PlyData([el]).write(path) produces)."""
    table = np.ascontiguousarray(table, dtype="<f4")
    assert table.ndim == 2 and table.shape[1] == len(names)
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {table.shape[0]}"]
    header += [f"property float {n}" for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(table.tobytes())


def read_vertices(path: str) -> Tuple[List[str], Dict[str, np.ndarray]]:
    """-> (property names in file order, {name: array[N]}) of the first `vertex` element."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex, seen_vertex = None, 0, [], False, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex" and not seen_vertex
                if in_vertex:
                    n, seen_vertex = int(tok[2]), True
                elif not seen_vertex:
                    raise ValueError(f"{path}: an element precedes `vertex` (unsupported)")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are unsupported")
                props.append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=n, ndmin=2) if n else np.zeros((0, len(props)))
            return names, {nm: rows[:, k].astype(ty) for k, (nm, ty) in enumerate(props)}
        if fmt not in ("binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unknown PLY format {fmt}")
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(nm, order + ty) for nm, ty in props])
        rec = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
        return names, {nm: np.ascontiguousarray(rec[nm]) for nm in names}
