import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
from raytracing_amd import api, scenes
ctx = api.Context(0)
for name, fn, kw in (("scene1", scenes.scene1, {}), ("scene2", scenes.scene2, {}), ("mesh64k", scenes.synthetic_mesh_scene, dict(n_lat=128, n_lon=256)), ("mesh1m", scenes.synthetic_mesh_scene, {})):
    import time
    t = time.time()
    b, cam = fn(**kw)
    arrays = b.arrays()
    t1 = time.time()
    sc = api.Scene(ctx, *arrays)
    print(name, sc.info(), "build arrays %.2fs scene_create %.2fs" % (t1 - t, time.time() - t1))
