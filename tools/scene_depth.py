import sys
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
from raytracing_amd import api, scenes, _lib as L
ctx = api.Context(0)
for flags in (0, L.SCENE_GPU_BVH):
    b, cam = scenes.synthetic_mesh_scene(None)
    sc = api.Scene(ctx, *b.arrays(), hit_rule=L.HIT_NEAREST, flags=flags)
    print(flags, sc.info())
