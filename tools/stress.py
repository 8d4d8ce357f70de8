"""Large and odd configurations through the library once (no oracle): several batches per render, deep paths
over a big mesh, a tiny batch cap, both hit rules. Prints segments and checks the sums are finite."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
from raytracing_amd import api, scenes, _lib as L

def run(ctx, recipe, w, h, spp, depth, rule, **kw):
    b, cam_args = recipe(width=w, height=h, **kw)
    sc = api.Scene(ctx, *b.arrays(), hit_rule=rule)
    cam = api.camera_new(*cam_args)
    acc = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda:0")
    t0 = time.perf_counter()
    sc.render_accumulate(cam, api.render_params(L.INT_RANDOM_WALK, depth, spp, seed=1), acc.data_ptr(), 0)
    st = sc.collect_stats(0)
    dt = time.perf_counter() - t0
    ok = bool(torch.isfinite(acc).all()) and float(acc.min()) >= 0.0 and float(acc.sum()) > 0.0
    print("%-28s %dx%d spp %d depth %d rule %d: %d batches, %.1f Mseg in %.3f s -> %.0f Mseg/s, sums ok %s" %
          (recipe.__name__, w, h, spp, depth, rule, st["batches"], st["segments"] / 1e6, dt, st["segments"] / dt / 1e6, ok), flush=True)
    assert ok
    sc.close()

ctx = api.Context(0)
run(ctx, scenes.scene2, 3840, 2160, 64, 10, L.HIT_FIRST_INDEX)            # 531 M paths: four batches
run(ctx, scenes.scene2, 3840, 2160, 64, 10, L.HIT_FIRST_INDEX)            # again, warm
run(ctx, scenes.synthetic_mesh_scene, 1920, 1080, 8, 32, L.HIT_NEAREST)   # deep chains over the 1 M-triangle mesh
run(ctx, scenes.synthetic_mesh_scene, 1920, 1080, 8, 32, L.HIT_FIRST_INDEX)
run(ctx, scenes.scene1, 1000, 563, 16, 10, L.HIT_FIRST_INDEX)
small = api.Context(0, max_batch_paths=1 << 16)
run(small, scenes.scene2, 640, 360, 16, 20, L.HIT_FIRST_INDEX)
print("stress ok")
