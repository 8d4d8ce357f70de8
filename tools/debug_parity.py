"""Scratch diagnostics run on the GPU box: prints where GPU and oracle disagree."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import _oracle as O  # noqa: E402
import _scenes as S  # noqa: E402
from raytracing_amd import api, scenes  # noqa: E402

ctx = api.Context(0)

# 1. mesh primary-ray mismatches
osc, ocam, _ = S.oracle_scene(S.small_mesh_scene)
gsc, gcam, _ = S.library_scene(ctx, S.small_mesh_scene)
rays = S.primary_rays(ocam)
g, r = gsc.intersect_batch(rays), osc.intersect_batch(rays)
bad = np.nonzero((g["obj"] != r["obj"]) | (g["prim"] != r["prim"]))[0]
print("mesh mismatches", len(bad))
for i in bad[:8]:
    print(i, "pixel", i % ocam.width, i // ocam.width, "ray", rays[i], "\n   gpu", g[i], "\n   ref", r[i])

# 2. albedo image, 1 spp
osc, ocam, _ = S.oracle_scene(scenes.scene1, width=64, height=40)
gsc, gcam, _ = S.library_scene(ctx, scenes.scene1, width=64, height=40)
for spp in (1, 2):
    ref8, ref_xyz, ref_lin, _ = osc.render(ocam, O.INT_ALBEDO, 0, spp, seed=9)
    gpu8, gpu_xyz, st = gsc.render(gcam, api.render_params(O.INT_ALBEDO, 0, spp, seed=9))
    diff = (gpu_xyz != ref_lin).any(axis=2)
    print("albedo spp", spp, "differing pixels", diff.sum(), "of", diff.size, st)
    ys, xs = np.nonzero(diff)
    for y, x in list(zip(ys, xs))[:6]:
        print("  (x=%d,y=%d) gpu %s ref %s" % (x, y, gpu_xyz[y, x], ref_lin[y, x]))
    print("  rows with diffs", np.unique(ys)[:20], "cols", np.unique(xs)[:20])
