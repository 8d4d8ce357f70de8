"""Glossy (Ns) and dielectric (Ni) materials: extensions with no counterpart in the reference, which parses
both numbers (src/material.rs:20-22,49-50) and never reads them. Their behaviour is defined by the oracle;
nothing in the reference pins it (parity unpinned). These tests fix the definition's properties on the CPU
and check the GPU kernels against the oracle."""
import ctypes as C
import math

import numpy as np
import pytest

import _oracle as O
import _scenes as S
from raytracing_amd import scenes


def lobe(axis, ns, r1, r2):
    out = (C.c_double * 3)()
    O.lib().ref_phong_lobe_direction(O.vec3(*axis), ns, r1, r2, out)
    return np.array(list(out))


def dielectric(d, nf, entering, ni, u):
    out = (C.c_double * 3)()
    reflected = O.lib().ref_dielectric_direction(O.vec3(*d), O.vec3(*nf), int(entering), ni, u, out)
    return bool(reflected), np.array(list(out))


def test_phong_lobe_is_unit_and_tightens_with_the_exponent(oracle):
    rng = np.random.default_rng(1)
    axis = np.array([0.3, -0.5, 0.81])
    axis /= np.linalg.norm(axis)
    for ns, expect in ((0.0, 1.0 / 2.0), (50.0, 51.0 / 52.0), (1000.0, 1001.0 / 1002.0)):
        cos = []
        for _ in range(4000):
            v = lobe(axis, ns, rng.random(), rng.random())
            assert abs(np.linalg.norm(v) - 1.0) < 1e-12
            cos.append(float(v @ axis))
        # pdf ~ cos^ns over the hemisphere about the axis: E[cos] = (ns + 1) / (ns + 2)
        assert abs(np.mean(cos) - expect) < 4.0 * np.std(cos) / math.sqrt(len(cos)) + 1e-4
        assert min(cos) >= 0.0
    assert np.allclose(lobe(axis, 1e9, 0.5, 0.3), axis, atol=1e-4)      # an infinitely sharp lobe is the mirror


def test_dielectric_directions(oracle):
    n = (0.0, 0.0, 1.0)
    # normal incidence from outside: Schlick gives R0 = 0.04 for ni = 1.5
    down = (0.0, 0.0, -1.0)
    assert dielectric(down, n, True, 1.5, 0.039)[0] and not dielectric(down, n, True, 1.5, 0.041)[0]
    refl, r = dielectric(down, n, True, 1.5, 0.0)
    assert np.allclose(r, (0.0, 0.0, 1.0))
    # Snell at 45 degrees
    d = np.array([math.sin(math.pi / 4), 0.0, -math.cos(math.pi / 4)])
    refl, t = dielectric(d, n, True, 1.5, 0.999)
    assert not refl
    t /= np.linalg.norm(t)
    assert abs(t[0] - math.sin(math.pi / 4) / 1.5) < 1e-12 and t[2] < 0
    # leaving the glass beyond the critical angle (41.8 degrees): always reflected
    inside = np.array([math.sin(math.radians(50)), 0.0, math.cos(math.radians(50))])
    refl, r = dielectric(inside, (0.0, 0.0, -1.0), False, 1.5, 0.999)
    assert refl and np.allclose(r, (inside[0], 0.0, -inside[2]))
    # index 1: straight through at normal incidence
    refl, t = dielectric(down, n, True, 1.0, 0.5)
    assert not refl and np.allclose(t, down)


def test_extended_scene_renders_on_the_oracle(oracle):
    osc, ocam, _ = S.oracle_scene(scenes.scene3_extended, width=96, height=54)
    img, _, lin, st = osc.render(ocam, O.INT_RANDOM_WALK, 32, 8, seed=2)
    assert img.any() and np.isfinite(lin).all()
    assert st.paths == 96 * 54 * 8 and st.paths < st.segments <= 32 * st.paths
    # the same seed gives the same image
    again, *_ = osc.render(ocam, O.INT_RANDOM_WALK, 32, 8, seed=2)
    assert (img == again).all()


@pytest.mark.gpu
@pytest.mark.parametrize("depth,spp", [(32, 4), (10, 8)])
def test_extended_scene_matches_oracle_on_gpu(depth, spp):
    """The only arithmetic that is not the same IEEE operation on both sides is pow / sincos (last bit), so
    nearly every path is bit-identical; the bar is looser than for the reference's own materials."""
    from raytracing_amd import api
    ctx = api.Context(0)
    osc, ocam, _ = S.oracle_scene(scenes.scene3_extended, width=192, height=108)
    gsc, gcam, _ = S.library_scene(ctx, scenes.scene3_extended, width=192, height=108)
    ref8, _, ref_lin, ref_st = osc.render(ocam, O.INT_RANDOM_WALK, depth, spp, seed=12)
    gpu8, gpu_xyz, st = gsc.render(gcam, api.render_params(O.INT_RANDOM_WALK, depth, spp, seed=12))
    assert st["paths"] == ref_st.paths
    assert abs(st["segments"] - ref_st.segments) <= 2e-3 * ref_st.segments
    assert (gpu_xyz == ref_lin).all(axis=2).mean() >= 0.99
    assert (gpu8 == ref8).mean() >= 0.99
    rays = S.primary_rays(ocam)
    ref, _ = osc.trace_batch(rays, O.INT_RANDOM_WALK, depth, seed=3)
    gpu, _ = gsc.trace_batch(rays, O.INT_RANDOM_WALK, depth, seed=3)
    assert (gpu.view(np.uint32) == ref.view(np.uint32)).all(axis=1).mean() >= 0.995
    ctx.close()
