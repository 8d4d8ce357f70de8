import sys
sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np
import _oracle as O, _scenes as S
from raytracing_amd import api, scenes
ctx = api.Context(0)
for name, recipe, kw in (("scene1", scenes.scene1, dict(width=200, height=112)), ("scene2", scenes.scene2, dict(width=212, height=120)),
                         ("scene3", scenes.scene3, dict(width=320, height=180)), ("multi", S.multi_mesh_scene, {})):
    osc, ocam, _ = S.oracle_scene(recipe, **kw)
    gsc, _, _ = S.library_scene(ctx, recipe, **kw)
    for rays in (S.primary_rays(ocam), S.random_rays(50000, 3)):
        g, r = gsc.intersect_batch(rays), osc.intersect_batch(rays)
        hit = (r["obj"] != O.MISS) & (g["obj"] == r["obj"])
        bt = (g["t"].view(np.uint64) == r["t"].view(np.uint64))[hit].mean()
        bp = (g["p"].view(np.uint64) == r["p"].view(np.uint64)).all(axis=1)[hit].mean()
        bn = (g["n"].view(np.uint64) == r["n"].view(np.uint64)).all(axis=1)[hit].mean()
        print(name, len(rays), "hits", int(hit.sum()), "bitwise t %.6f p %.6f n %.6f" % (bt, bp, bn))
