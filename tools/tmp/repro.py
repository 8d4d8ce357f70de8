import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
from raytracing_amd import api, scenes, _lib as L
ctx = api.Context(0, 0)
for name, recipe, kw, spp in (("suzanne", "scene2", dict(width=1920, height=1080), 8), ("scene1", "scene1", dict(width=1600, height=900), 8),
                              ("mesh", "synthetic_mesh_scene", dict(n_lat=64, n_lon=128, width=1920, height=1080), 8)):
    b, cam_args = getattr(scenes, recipe)(**kw)
    sc = api.Scene(ctx, *b.arrays(), hit_rule=L.HIT_FIRST_INDEX)
    cam = api.camera_new(*cam_args)
    stream = torch.cuda.current_stream().cuda_stream
    outs = []
    for rep in range(3):
        acc = torch.zeros(cam.height * cam.width * 3, dtype=torch.float32, device="cuda:0")
        sc.render_accumulate(cam, api.render_params(L.INT_RANDOM_WALK, 10, spp, seed=0), acc.data_ptr(), stream)
        st = sc.collect_stats(stream)
        torch.cuda.synchronize()
        outs.append((acc, st["segments"]))
    d1 = int((outs[0][0].view(torch.int32) != outs[1][0].view(torch.int32)).sum()); d2 = int((outs[0][0].view(torch.int32) != outs[2][0].view(torch.int32)).sum())
    print(name, "lanes", os.environ.get("RTMI_LANES", "2"), "differing floats", d1, d2, "segments", [o[1] for o in outs], flush=True)
