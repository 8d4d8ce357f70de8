#!/usr/bin/env python3
"""Extracts the geometry of the reference's two mesh assets into small binary fixtures.

Reads /root/reference/{cubo,mono}.obj (positions and polygon faces only -- the reference's
loader ignores normals and texture coordinates, src/shapes/model_obj.rs:40-53) and writes
raytracing_amd/data/meshes.npz (the package ships it: the benchmark scenes use it) with, per mesh, the vertex array (f64) and the polygon faces as a flat
0-based index array plus per-face offsets. Polygons are kept as polygons so that either fan
triangulation can be derived from the fixture. Run once in the build container; the GPU box
has no /root/reference and only ever reads the .npz.
"""
import sys
from pathlib import Path

import numpy as np

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent.parent.parent / "raytracing_amd" / "data" / "meshes.npz"


def read_obj(path):
    verts, idx, offs = [], [], [0]
    for line in path.read_text().splitlines():
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "v":
            verts.append([float(t) for t in tok[1:4]])
        elif tok[0] == "f":
            for corner in tok[1:]:
                i = int(corner.split("/")[0])
                idx.append(i - 1 if i > 0 else len(verts) + i)
            offs.append(len(idx))
    return np.array(verts, dtype=np.float64), np.array(idx, dtype=np.int32), np.array(offs, dtype=np.int32)


def main():
    out = {}
    for name, fname in (("cubo", "cubo.obj"), ("mono", "mono.obj")):
        v, i, o = read_obj(REF / fname)
        out[name + "_vertices"], out[name + "_face_index"], out[name + "_face_offsets"] = v, i, o
        print(name, v.shape, len(o) - 1, "faces")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
