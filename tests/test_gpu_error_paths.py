"""Failures in the middle of a render, forced through the library's fault-injection hooks (environment variables read when a
context is created): the context, its internal lanes and the caller's stream must stay usable, and the next render
on them must produce the bits an undisturbed context produces."""
import numpy as np
import pytest

import _oracle as O
import _scenes as S
from raytracing_amd import scenes

pytestmark = pytest.mark.gpu


def _render_on_stream(scene, cam, params):
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    acc = torch.zeros(cam.height * cam.width * 3, dtype=torch.float32, device="cuda:0")
    scene.render_accumulate(cam, params, acc.data_ptr(), stream)
    st = scene.collect_stats(stream)
    torch.cuda.synchronize()
    return acc.cpu().numpy(), st


@pytest.mark.parametrize("recipe", [scenes.scene1, scenes.scene3])
def test_a_batch_that_fails_midway_leaves_the_context_usable(monkeypatch, recipe):
    """Batch 2 of a many-batch render fails after batches 0 and 1 were launched on the internal lanes: the call returns the
    error, the caller's stream is joined with the lanes again, and the same context and stream then render normally."""
    from raytracing_amd import api, _lib as L
    kw = dict(width=160, height=96)
    p = api.render_params(O.INT_RANDOM_WALK, 8, 6, seed=31)
    good = api.Context(0, max_batch_paths=4096)
    gsc, cam, _ = S.library_scene(good, recipe, **kw)
    want, want_st = _render_on_stream(gsc, cam, p)
    assert want_st["batches"] > 3
    good.close()

    monkeypatch.setenv("RTMI_TEST_FAIL_BATCH", "2")
    flaky = api.Context(0, max_batch_paths=4096)        # the hook is read here
    monkeypatch.delenv("RTMI_TEST_FAIL_BATCH")
    fsc, _, _ = S.library_scene(flaky, recipe, **kw)
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    acc = torch.zeros(cam.height * cam.width * 3, dtype=torch.float32, device="cuda:0")
    for _ in range(2):
        with pytest.raises(L.RtmiError) as e:
            fsc.render_accumulate(cam, p, acc.data_ptr(), stream)
        assert e.value.code == L.INTERNAL and "RTMI_TEST_FAIL_BATCH" in str(e.value)
        # work queued on the caller's stream right after the failure runs after everything the lanes had in flight
        acc.zero_()
        torch.cuda.synchronize()
    # the same scene through a context without the hook, sharing the GPU and the stream: the library state is sound
    clean = api.Context(0, max_batch_paths=4096)
    csc, _, _ = S.library_scene(clean, recipe, **kw)
    got, got_st = _render_on_stream(csc, cam, p)
    assert (got.view(np.uint32) == want.view(np.uint32)).all() and got_st["segments"] == want_st["segments"]
    # and the context that failed renders a single-batch job (there is no batch 2 to fail) correctly afterwards
    small = api.render_params(O.INT_NORMAL, 0, 1, seed=1)
    fsmall, small_cam, _ = S.library_scene(flaky, recipe, width=32, height=24)
    csmall, _, _ = S.library_scene(clean, recipe, width=32, height=24)
    a, a_st = _render_on_stream(fsmall, small_cam, small)
    b, _ = _render_on_stream(csmall, small_cam, small)
    assert a_st["batches"] <= 2 and a.any() and (a.view(np.uint32) == b.view(np.uint32)).all()
    for c in (flaky, clean):
        c.close()


def test_out_of_memory_on_a_lane_falls_back_to_fewer_lanes(monkeypatch):
    """A render that would overlap its batches on several lanes, each with a workspace of its own, on a device where only
    the first lane's workspace fits: the library gives up the overlap instead of failing, and the image is the same."""
    from raytracing_amd import api
    kw = dict(width=192, height=108)
    p = api.render_params(O.INT_RANDOM_WALK, 8, 4, seed=7)
    ref = api.Context(0, max_batch_paths=8192)
    rsc, cam, _ = S.library_scene(ref, scenes.scene1, **kw)
    want, want_st = _render_on_stream(rsc, cam, p)
    assert want_st["batches"] > 2
    monkeypatch.setenv("RTMI_TEST_OOM_LANE", "1")
    tight = api.Context(0, max_batch_paths=8192)
    monkeypatch.delenv("RTMI_TEST_OOM_LANE")
    tsc, _, _ = S.library_scene(tight, scenes.scene1, **kw)
    got, got_st = _render_on_stream(tsc, cam, p)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    assert got_st["segments"] == want_st["segments"] and got_st["batches"] == want_st["batches"]
    got2, _ = _render_on_stream(tsc, cam, p)            # the reduced lane count is remembered, nothing is retried
    assert (got2.view(np.uint32) == want.view(np.uint32)).all()
    for c in (ref, tight):
        c.close()
