import sys, numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import _oracle as O, _scenes as S
from raytracing_amd import api
ctx = api.Context(0)
for rule in (0, 1):
    osc, ocam, _ = S.oracle_scene(S.small_mesh_scene, hit_rule=rule)
    gsc, _, _ = S.library_scene(ctx, S.small_mesh_scene, hit_rule=rule)
    rays = S.primary_rays(ocam)
    g, r = gsc.intersect_batch(rays), osc.intersect_batch(rays)
    bad = np.nonzero((g["obj"] != r["obj"]) | (g["prim"] != r["prim"]))[0]
    print("rule", rule, "bad", len(bad), gsc.info())
    for i in bad[:12]:
        print(i, "gpu", g["obj"][i], g["prim"][i], g["t"][i], "ref", r["obj"][i], r["prim"][i], r["t"][i], "d", rays["d"][i])
