"""Per-bounce segment / node / triangle-test counts for a workload (cumulative renders at depth 1..D)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
from raytracing_amd import api, scenes, _lib as L
import torch
ctx = api.Context(0)
rule = L.HIT_NEAREST if len(sys.argv) > 1 and sys.argv[1] == "nearest" else L.HIT_FIRST_INDEX
b, cam_args = scenes.scene2(width=1920, height=1080)
sc = api.Scene(ctx, *b.arrays(), hit_rule=rule)
cam = api.camera_new(*cam_args)
acc = torch.zeros(1920 * 1080 * 3, dtype=torch.float32, device="cuda:0")
prev = dict(segments=0, nodes_visited=0, prims_tested=0, mesh_rays=0)
for depth in range(1, 11):
    acc.zero_()
    sc.render_accumulate(cam, api.render_params(L.INT_RANDOM_WALK, depth, 16, flags=L.FLAG_COUNT_TRAVERSAL), acc.data_ptr(), 0)
    st = sc.collect_stats(0)
    d = {k: st[k] - prev[k] for k in prev}
    print("bounce %d: segments %9d mesh rays %9d nodes/ray %7.2f tris/ray %7.2f" % (depth - 1, d["segments"], d["mesh_rays"], d["nodes_visited"] / max(1, d["mesh_rays"]), d["prims_tested"] / max(1, d["mesh_rays"])))
    prev = {k: st[k] for k in prev}
