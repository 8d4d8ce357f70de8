"""Paths alive at each bounce level of a bench workload: renders at max_depth 1..D and differences the segment counts.
   python tools/level_populations.py [workload] [spp] [rule]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracing_amd import api, _lib as L

name = sys.argv[1] if len(sys.argv) > 1 else "suzanne_1080p_64spp"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rule = sys.argv[3] if len(sys.argv) > 3 else "first_index"
w = bench.WORKLOADS[name]
ctx = api.Context(0)
sc, cam = bench.build_scene(ctx, w, L.HIT_FIRST_INDEX if rule == "first_index" else L.HIT_NEAREST)
prev = {}
for depth in range(1, w["depth"] + 1):
    row = {}
    st = sc.render(cam, api.render_params(0, depth, spp, seed=1, flags=L.FLAG_COUNT_TRAVERSAL), want_xyz=False)[-1]
    for k in ("segments", "mesh_rays", "nodes_visited", "prims_tested"):
        row[k] = int(st[k])
    st = sc.render(cam, api.render_params(0, depth, spp, seed=1, flags=L.FLAG_TIME_KERNELS), want_xyz=False)[-1]
    row["ms_traverse"], row["ms_kernels"] = float(st["ms_traverse"]), float(st["ms_kernels"])
    d = {k: row[k] - prev.get(k, 0) for k in row}
    rays = max(1, d["mesh_rays"])
    print("level %2d  segments %11d  mesh rays %10d  node steps/ray %6.1f  triangle tests/ray %6.1f  traverse %7.3f ms (%6.2f rays/us)  all kernels %7.3f ms"
          % (depth - 1, d["segments"], d["mesh_rays"], d["nodes_visited"] / rays, d["prims_tested"] / rays, d["ms_traverse"],
             d["mesh_rays"] / max(1e-9, d["ms_traverse"]) / 1e3, d["ms_kernels"]), flush=True)
    prev = row
