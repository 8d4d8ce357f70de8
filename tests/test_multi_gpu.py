"""rtmi_multi_*: one render partitioned over several devices inside one process, merged by one collective.

The GPU box has one MI355X, so the device list repeats device 0 (shares that live on one GPU are merged by copies and
adds) and the RCCL path is exercised with a one-rank communicator (RTMI_RCCL_SINGLE=1); the N-process form of the same
partition runs over gloo in tests/test_parallel_gloo.py."""
import numpy as np
import pytest

import _oracle as O
import _scenes as S
from raytracing_amd import scenes

pytestmark = pytest.mark.gpu


def _single(recipe, params, **kw):
    from raytracing_amd import api
    ctx = api.Context(0, max_batch_paths=1 << 22)
    sc, cam, _ = S.library_scene(ctx, recipe, **kw)
    out = sc.render(cam, params)
    ctx.close()
    return cam, out


@pytest.mark.parametrize("recipe,kw", [(scenes.scene1, dict(width=200, height=112)), (scenes.scene3, dict(width=160, height=90))])
def test_multi_render_matches_single_device(recipe, kw):
    from raytracing_amd import api, _lib as L
    p = api.render_params(O.INT_RANDOM_WALK, 10, 7, seed=12)
    cam, (ref8, ref_xyz, ref_st) = _single(recipe, p, **kw)
    b, _ = recipe(from_rgb=None, **kw)
    # one rank: no collective, the same bits as rtmi_render
    one = api.MultiScene([0], *b.arrays(), max_batch_paths=1 << 22)
    assert one.info() == {"n_ranks": 1, "uses_rccl": False, "rccl_library": ""}
    for part in (L.PARTITION_SAMPLES, L.PARTITION_TILES):
        img, xyz, st, per = one.render(cam, p, part)
        assert (img == ref8).all() and (xyz.view(np.uint32) == ref_xyz.view(np.uint32)).all()
        assert st["segments"] == ref_st["segments"] and st["paths"] == ref_st["paths"] and len(per) == 1
    one.close()
    # three shares on one GPU: tiles are disjoint, so the merged image is bit-identical; sample ranges are summed in
    # another order (a + b + c of partial sums instead of one running sum), which moves a pixel by float rounding
    three = api.MultiScene([0, 0, 0], *b.arrays(), max_batch_paths=1 << 22)
    assert three.info()["n_ranks"] == 3 and not three.info()["uses_rccl"]
    img, xyz, st, per = three.render(cam, p, L.PARTITION_TILES)
    assert (img == ref8).all() and (xyz.view(np.uint32) == ref_xyz.view(np.uint32)).all()
    assert st["segments"] == ref_st["segments"] and sum(s["paths"] for s in per) == ref_st["paths"]
    assert all(s["paths"] > 0 for s in per)
    img, xyz, st, per = three.render(cam, p, L.PARTITION_SAMPLES)
    assert st["segments"] == ref_st["segments"] and [s["paths"] for s in per] == [cam.width * cam.height * k for k in (3, 2, 2)]
    assert np.abs(xyz - ref_xyz).max() <= 4e-7 * max(1.0, float(np.abs(ref_xyz).max()))
    assert (img == ref8).mean() >= 0.999
    # more ranks than samples: the ranks without a sample contribute zeros
    p1 = api.render_params(O.INT_NORMAL, 0, 2, seed=3)
    _, (n8, nxyz, _) = _single(recipe, p1, **kw)
    img, xyz, st, per = three.render(cam, p1, L.PARTITION_SAMPLES)
    assert [s["paths"] for s in per][2] == 0 and np.abs(xyz - nxyz).max() <= 4e-7 and (img == n8).mean() >= 0.999
    # the partition fields belong to the library here
    with pytest.raises(L.RtmiError) as e:
        three.render(cam, api.render_params(O.INT_RANDOM_WALK, 10, 4, tile_rank=0, tile_world=2), L.PARTITION_TILES)
    assert e.value.code == L.BAD_ARG
    three.close()


def test_rccl_binding_with_one_rank(monkeypatch):
    """ncclCommInitAll + ncclReduce through the run-time binding, on the one device there is: a one-rank reduce leaves
    the sums as they are, so the image must still be rtmi_render's."""
    from raytracing_amd import api, _lib as L
    p = api.render_params(O.INT_RANDOM_WALK, 10, 4, seed=5)
    cam, (ref8, ref_xyz, ref_st) = _single(scenes.scene1, p, width=128, height=72)
    b, _ = scenes.scene1(from_rgb=None, width=128, height=72)
    monkeypatch.setenv("RTMI_RCCL_SINGLE", "1")
    m = api.MultiScene([0], *b.arrays())
    info = m.info()
    assert info["uses_rccl"] and "rccl" in info["rccl_library"]
    img, xyz, st, _ = m.render(cam, p, L.PARTITION_SAMPLES)
    assert (img == ref8).all() and (xyz.view(np.uint32) == ref_xyz.view(np.uint32)).all() and st["segments"] == ref_st["segments"]
    m.close()


def test_progressive_passes_add_up_to_the_single_render():
    """k passes of spp / k samples: every pass returns the image of the samples so far (= a render of that many
    samples), and the last one the full render -- bit for bit under the tile partition."""
    from raytracing_amd import api, _lib as L
    kw = dict(width=160, height=90)
    b, _ = scenes.scene1(from_rgb=None, **kw)
    spp = 7
    cam, (full8, full_xyz, full_st) = _single(scenes.scene1, api.render_params(O.INT_RANDOM_WALK, 10, spp, seed=2), **kw)
    _, (first8, first_xyz, _) = _single(scenes.scene1, api.render_params(O.INT_RANDOM_WALK, 10, 3, seed=2), **kw)
    m = api.MultiScene([0, 0], *b.arrays(), max_batch_paths=1 << 22)
    p = api.render_params(O.INT_RANDOM_WALK, 10, spp, seed=2)
    segments = 0
    for k in range(3):          # 3 + 2 + 2 samples
        img, xyz, st = m.render_pass(cam, p, L.PARTITION_TILES, k, 3)
        segments += st["segments"]
        if k == 0:              # the first pass is the image of samples 0..2
            assert (img == first8).all() and (xyz.view(np.uint32) == first_xyz.view(np.uint32)).all()
    assert (img == full8).all() and (xyz.view(np.uint32) == full_xyz.view(np.uint32)).all() and segments == full_st["segments"]
    for k in range(2):
        img, xyz, st = m.render_pass(cam, p, L.PARTITION_SAMPLES, k, 2)
    assert np.abs(xyz - full_xyz).max() <= 4e-7 * max(1.0, float(np.abs(full_xyz).max())) and (img == full8).mean() >= 0.999
    with pytest.raises(L.RtmiError):
        m.render_pass(cam, p, L.PARTITION_TILES, 3, 3)
    m.close()


def test_passes_out_of_order_are_refused():
    """The ranks' sums carry over between passes, so a pass that does not continue the render in flight -- a fresh handle
    asked for pass 1, a pass skipped or repeated, another camera, sample count or partition half way -- must be refused
    (RTMI_BAD_ARG) instead of adding to sums that belong to another image; and a refused pass must not disturb the
    render in flight."""
    from raytracing_amd import api, _lib as L
    kw = dict(width=96, height=54)
    b, _ = scenes.scene3(from_rgb=None, **kw)
    p = api.render_params(O.INT_RANDOM_WALK, 6, 6, seed=4)
    cam, (full8, full_xyz, _) = _single(scenes.scene3, p, **kw)
    m = api.MultiScene([0, 0], *b.arrays(), max_batch_paths=1 << 22)

    def refused(*args):
        with pytest.raises(L.RtmiError) as e:
            m.render_pass(*args)
        assert e.value.code == L.BAD_ARG

    refused(cam, p, L.PARTITION_TILES, 1, 3)                      # nothing in flight
    m.render_pass(cam, p, L.PARTITION_TILES, 0, 3)
    refused(cam, p, L.PARTITION_TILES, 2, 3)                      # skips pass 1
    refused(cam, p, L.PARTITION_TILES, 1, 2)                      # another pass count
    refused(cam, p, L.PARTITION_SAMPLES, 1, 3)                    # another partition
    refused(cam, api.render_params(O.INT_RANDOM_WALK, 6, 9, seed=4), L.PARTITION_TILES, 1, 3)    # another sample count
    other = api.camera_new((-4.5, 0.0, 2.0), 0.1, 120.0, (0.0, 0.5, 0.0), 64, 36)
    refused(other, p, L.PARTITION_TILES, 1, 3)                    # another resolution
    m.render_pass(cam, p, L.PARTITION_TILES, 1, 3)
    refused(cam, p, L.PARTITION_TILES, 1, 3)                      # repeats pass 1
    img, xyz, _ = m.render_pass(cam, p, L.PARTITION_TILES, 2, 3)
    assert (img == full8).all() and (xyz.view(np.uint32) == full_xyz.view(np.uint32)).all()
    # a whole render ends the progressive one
    m.render(cam, p, L.PARTITION_TILES)
    refused(cam, p, L.PARTITION_TILES, 1, 3)
    m.close()
