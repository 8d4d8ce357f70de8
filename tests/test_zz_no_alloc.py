"""rtmi_render_accumulate_noalloc (static scratch array, no device allocation) against rtmi_render_accumulate.

Both run the same kernels over the same paths; only where the workspace lives and how the render is cut into batches
differ, and the image does not depend on the batches. So the per-pixel sums must agree bit for bit, on two calls in a
row, the second on a smaller input (the static array keeps the first call's contents).
"""
import numpy as np
import pytest

import _oracle as O
import _scenes as S
from raytracing_amd import scenes

pytestmark = pytest.mark.gpu


def _pair(scene, cam, params, torch, stream):
    w, h = cam.width, cam.height
    a = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
    b = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
    scene.render_accumulate(cam, params, a.data_ptr(), stream)
    sa = scene.collect_stats(stream)
    scene.render_accumulate_noalloc(cam, params, b.data_ptr(), stream)
    sb = scene.collect_stats(stream)
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy(), sa, sb


def test_noalloc_render_equals_allocating_render():
    import torch
    from raytracing_amd import api, _lib as L
    ctx = api.Context(0, max_batch_paths=1 << 22)
    stream = torch.cuda.current_stream().cuda_stream
    # first call: the cube room (a mesh: two-kernel pipeline), 400 x 225 x 6 spp = 540 000 paths = 5 batches of the
    # static workspace; second call, smaller: 96 x 54 x 3 spp with a tile partition
    big, bcam, _ = S.library_scene(ctx, scenes.scene1, width=400, height=225)
    a, b, sa, sb = _pair(big, bcam, api.render_params(O.INT_RANDOM_WALK, 10, 6, seed=3), torch, stream)
    assert a.any() and (a.view(np.uint32) == b.view(np.uint32)).all()
    assert sa["segments"] == sb["segments"] and sa["paths"] == sb["paths"] and sb["batches"] > sa["batches"]
    small, scam, _ = S.library_scene(ctx, scenes.scene1, width=96, height=54)
    a, b, sa, sb = _pair(small, scam, api.render_params(O.INT_RANDOM_WALK, 10, 3, seed=4, tile_rank=1, tile_world=2), torch, stream)
    assert a.any() and (a.view(np.uint32) == b.view(np.uint32)).all()
    assert sa["segments"] == sb["segments"] and sa["paths"] == sb["paths"]
    # a scene without meshes (fused kernel), depth 20 (the long-chain queues), and the AOV integrators
    s3, cam3, _ = S.library_scene(ctx, scenes.scene3, width=160, height=90)
    for params in (api.render_params(O.INT_RANDOM_WALK, 20, 4, seed=5, sample_range=(1, 3)), api.render_params(O.INT_NORMAL, 0, 2, seed=6),
                   api.render_params(O.INT_ALBEDO, 0, 2, seed=6)):
        a, b, sa, sb = _pair(s3, cam3, params, torch, stream)
        assert a.any() and (a.view(np.uint32) == b.view(np.uint32)).all() and sa["segments"] == sb["segments"]
    # and the result is the oracle's
    osc, ocam, _ = S.oracle_scene(scenes.scene1, width=96, height=54)
    _, _, ref_lin, _ = osc.render(ocam, O.INT_RANDOM_WALK, 10, 3, seed=4)
    acc = torch.zeros(54 * 96 * 3, dtype=torch.float32, device="cuda:0")
    small.render_accumulate_noalloc(scam, api.render_params(O.INT_RANDOM_WALK, 10, 3, seed=4), acc.data_ptr(), stream)
    torch.cuda.synchronize()
    mean = (acc.cpu().numpy() / np.float32(3)).reshape(54, 96, 3)
    assert (mean == ref_lin).all(axis=2).mean() >= 0.999
    # sizes beyond the static workspace are refused with a clear error
    with pytest.raises(L.RtmiError) as e:
        small.render_accumulate_noalloc(scam, api.render_params(O.INT_RANDOM_WALK, 10, 1, tile_size=512), acc.data_ptr(), stream)
    assert e.value.code == L.UNSUPPORTED and "tile" in str(e.value)
    with pytest.raises(L.RtmiError) as e:
        small.render_accumulate_noalloc(scam, api.render_params(O.INT_RANDOM_WALK, 10, 1, flags=L.FLAG_TIME_KERNELS), acc.data_ptr(), stream)
    assert e.value.code == L.UNSUPPORTED
    ctx.close()
