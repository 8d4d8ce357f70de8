"""Behaviours of the reference that none of its own tests pin, restated as tests of the oracle.

Each test names the reference lines whose behaviour it fixes. These are the facts the HIP
kernels are then held to by tests/test_gpu_parity.py.
"""
import ctypes as C

import numpy as np
import pytest

from raytracing_amd.scene_builder import SceneBuilder


def rays(o_list, d_list, O):
    r = np.zeros(len(o_list), dtype=O.RAY_DT)
    for k, (o, d) in enumerate(zip(o_list, d_list)):
        q = O.make_ray(o, d)
        r["o"][k], r["d"][k] = list(q.o), list(q.d)
    return r


def test_mesh_returns_first_triangle_in_file_order_not_nearest(oracle):
    """src/shapes/model_obj.rs:80-84: two stacked triangles, the FAR one listed first."""
    O = oracle
    tri_far = [0, 0, 5, 1, 0, 5, 0, 1, 5]
    tri_near = [0, 0, 2, 1, 0, 2, 0, 1, 2]
    for rule, want_prim, want_t in ((O.HIT_FIRST_INDEX, 0, 6.0), (O.HIT_NEAREST, 1, 3.0)):
        b = SceneBuilder(O.from_rgb)
        b.add_model(np.array([tri_far, tri_near], dtype=np.float64))
        h = O.Scene(*b.arrays(), hit_rule=rule).intersect_batch(rays([(0.2, 0.2, -1)], [(0, 0, 1)], O))
        assert h["obj"][0] == 0 and h["prim"][0] == want_prim and abs(h["t"][0] - want_t) < 1e-12


def test_scene_keeps_the_earlier_object_on_a_tie(oracle):
    """src/scene.rs:141-147: strict '<' when reducing over objects."""
    O = oracle
    b = SceneBuilder(O.from_rgb)
    b.add_triangle([0, 0, 2, 1, 0, 2, 0, 1, 2])
    b.add_triangle([0, 0, 2, 1, 0, 2, 0, 1, 2])
    b.add_sphere((0.2, 0.2, 1.0), 0.5)
    h = O.Scene(*b.arrays()).intersect_batch(rays([(0.2, 0.2, 10.0), (0.2, 0.2, 3.0)], [(0, 0, -1), (0, 0, -1)], O))
    assert h["obj"][0] == 0 and h["obj"][1] == 0      # the two coincident triangles tie: the first wins
    h2 = O.Scene(*b.arrays()).intersect_batch(rays([(0.2, 0.2, -3.0)], [(0, 0, 1)], O))
    assert h2["obj"][0] == 2                           # from below the sphere is closer


def test_sphere_t_is_measured_along_the_local_ray(oracle):
    """src/shapes/sphere.rs:52,101 + src/geometry/ray.rs:45: a sphere scaled by 1/2 reports t in local
    units (twice the world distance), which can reorder it against other objects (src/scene.rs:142)."""
    O = oracle
    b = SceneBuilder(O.from_rgb)
    b.add_sphere((0, 0, 0), 1.0, scale=(0.5, 0.5, 0.5))      # world radius 0.5, hit at world distance 2.5
    b.add_triangle([-5, -5, -2, 5, -5, -2, 0, 5, -2])          # plane z = -2 behind it, world distance 5
    h = O.Scene(*b.arrays()).intersect_batch(rays([(0, 0, 3.0)], [(0, 0, -1)], O))
    assert h["obj"][0] == 0 and abs(h["t"][0] - 5.0) < 1e-12   # (3 - 0.5) / 0.5 = 5 in local units
    assert np.allclose(h["p"][0], (0, 0, 0.5), atol=1e-12)
    b2 = SceneBuilder(O.from_rgb)
    b2.add_sphere((0, 0, 0), 1.0, scale=(0.5, 0.5, 0.5))
    b2.add_triangle([-5, -5, -1.9, 5, -5, -1.9, 0, 5, -1.9])   # world distance 4.9 < local t 5: the plane "wins"
    h = O.Scene(*b2.arrays()).intersect_batch(rays([(0, 0, 3.0)], [(0, 0, -1)], O))
    assert h["obj"][0] == 1


def test_mesh_box_starts_from_the_origin_point(oracle):
    """src/geometry/aabb.rs:15-20 + src/shapes/model_obj.rs:62-66: AABB::empty() is the point (0,0,0)."""
    O = oracle
    b = SceneBuilder(O.from_rgb)
    b.add_model(np.array([[5, 5, 5, 6, 5, 5, 5, 6, 5]], dtype=np.float64))
    s = O.Scene(*b.arrays())
    lo, hi = O.vec3(0, 0, 0), O.vec3(0, 0, 0)
    O.lib().ref_scene_model_bounds(s.handle, 0, lo, hi)
    assert list(lo) == [0, 0, 0] and list(hi) == [6, 6, 5]


def two_sided_scene(O):
    """A Lambertian triangle seen from its back, with an emitter behind it (on the normal's side)."""
    b = SceneBuilder(O.from_rgb)
    # winding gives n = +z; the camera ray comes from z = +3 travelling -z?? no: from below (z<0) travelling +z, so n.d > 0
    b.add_triangle([-10, -10, 0, 10, -10, 0, 0, 10, 0], material=b.material("Lambertian", (0.5, 0.5, 0.5)))
    b.add_sphere((0, 0, 5.0), 3.0, material=b.material("Emitter", (1.0, 1.0, 1.0)))
    return b


def test_primary_hit_keeps_its_normal_secondary_hits_turn_it(oracle):
    """src/integrators/random_walk.rs:74 vs src/scene.rs:30-32. A primary ray that reaches a Lambertian
    triangle from behind scatters about the UNturned normal, i.e. through the surface, and can reach an
    emitter on the far side; the same surface reached by a bounce ray scatters back."""
    O = oracle
    s = O.Scene(*two_sided_scene(O).arrays())
    r = rays([(0.1, 0.1, -2.0)] * 64, [(0, 0, 1)] * 64, O)
    spectra, segs = s.trace_batch(r, O.INT_RANDOM_WALK, 2, seed=3)
    assert (spectra.sum(axis=1) > 0).mean() > 0.2          # light arrives through the back face
    assert segs == 128
    # the mirrored situation for a bounce: camera above, first hit on the front face, emitter below is unreachable
    b = SceneBuilder(O.from_rgb)
    b.add_triangle([-10, -10, 0, 10, -10, 0, 0, 10, 0], material=b.material("Lambertian", (0.5, 0.5, 0.5)))
    b.add_sphere((0, 0, -5.0), 3.0, material=b.material("Emitter", (1.0, 1.0, 1.0)))
    s2 = O.Scene(*b.arrays())
    spectra2, _ = s2.trace_batch(rays([(0.1, 0.1, 2.0)] * 64, [(0, 0, -1)] * 64, O), O.INT_RANDOM_WALK, 3, seed=3)
    assert not spectra2.any()


def test_depth_accounting(oracle):
    """src/integrators/random_walk.rs:52-70 and src/scene.rs:24-26: at most D ray casts per path; depth 0
    still casts the primary ray and returns black; trace_ray(.., 0) returns before intersecting."""
    O = oracle
    b = SceneBuilder(O.from_rgb)
    b.add_sphere((0, 0, 0), 1.0, material=b.material("Specular", (0.9, 0.9, 0.9)))
    b.add_sphere((0, 0, 4), 1.0, material=b.material("Specular", (0.9, 0.9, 0.9)))
    s = O.Scene(*b.arrays())
    r = rays([(0, 0, 2)], [(0, 0, -1)], O)       # bounces forever between the two mirrors
    for depth in (0, 1, 2, 5, 10):
        spectra, segs = s.trace_batch(r, O.INT_RANDOM_WALK, depth, seed=1)
        assert segs == max(1, depth) and not spectra.any()


def test_emitter_seen_through_a_tinted_mirror(oracle):
    """src/scene.rs:80-112 (perfect mirror, tint = specular colour, no Fresnel) and :56-62; the product is
    formed innermost first: (Le * c_mirror)."""
    O = oracle
    b = SceneBuilder(O.from_rgb)
    mirror = b.material("Specular", (0.2, 0.1, 0.9))
    light = b.material("Emitter", (0.6, 0.6, 0.6))
    b.add_triangle([-10, -10, 0, 10, -10, 0, 0, 10, 0], material=mirror)
    b.add_sphere((0, 1.5, 5), 1.0, material=light)
    s = O.Scene(*b.arrays())
    d = np.array([0.0, 0.3, -1.0])
    spectra, segs = s.trace_batch(rays([(0, -0.9, 3.0)], [tuple(d)], O), O.INT_RANDOM_WALK, 4, seed=1)
    expect = O.from_rgb((0.6, 0.6, 0.6), O.SPEC_ILLUMINANT) * O.from_rgb((0.2, 0.1, 0.9), O.SPEC_REFLECTANCE)
    assert segs == 2 and (spectra[0].view(np.uint32) == expect.view(np.uint32)).all()


def test_black_materials(oracle):
    """src/material.rs:25-37 (default = emitter without colour) and src/scene.rs:72-78 (Lambertian without
    ambient colour): both end the path with zero radiance."""
    O = oracle
    b = SceneBuilder(O.from_rgb)
    b.add_sphere((0, 0, 0), 1.0)
    b.add_sphere((3, 0, 0), 1.0, material=b._material(O.MAT_LAMBERTIAN))
    s = O.Scene(*b.arrays())
    spectra, segs = s.trace_batch(rays([(0, 0, 5), (3, 0, 5)], [(0, 0, -1), (0, 0, -1)], O), O.INT_RANDOM_WALK, 10, seed=1)
    assert not spectra.any() and segs == 2


def test_aov_integrators(oracle):
    """src/integrators/albedo.rs:54-68 (first present colour) and normal.rs:48-59."""
    O = oracle
    b = SceneBuilder(O.from_rgb)
    b.add_sphere((0, 0, 0), 1.0, material=b.material("Specular", (0.2, 0.4, 0.6)))
    b.add_sphere((3, 0, 0), 1.0, material=b.mtl_material((0.1, 0.2, 0.3), (0.9, 0.9, 0.9), (0.5, 0.5, 0.5), (0.7, 0.7, 0.7)))
    s = O.Scene(*b.arrays())
    r = rays([(0, 0, 5), (3, 0, 5), (9, 9, 5)], [(0, 0, -1)] * 3, O)
    albedo, _ = s.trace_batch(r, O.INT_ALBEDO, 0, seed=1)
    assert (albedo[0] == O.from_rgb((0.2, 0.4, 0.6))).all()      # only a specular colour
    assert (albedo[1] == O.from_rgb((0.1, 0.2, 0.3))).all()      # ambient (Ka) comes first
    assert not albedo[2].any()
    normal, _ = s.trace_batch(r, O.INT_NORMAL, 0, seed=1)
    assert (normal[0] == O.from_rgb((0.5, 0.5, 1.0))).all() and not normal[2].any()


def test_white_reflectance_gains_energy(oracle):
    """src/spectrum/data.rs:327-339: the Smits white basis is about 1.06, so Ka = 1 walls amplify."""
    w = oracle.from_rgb((1.0, 1.0, 1.0))
    assert 1.055 < w.min() and w.max() < 1.065


def test_white_to_rgb_and_y_normalisation(oracle):
    """A flat spectrum has X = Y = Z, so white does not come back as (1,1,1): each channel is the flat
    level times the row sum of the XYZ->RGB matrix (src/spectrum/mod.rs:630-636). Y of a unit spectrum
    is 1 by the scale 300 / (CIE_Y_INTEGRAL * 60) (src/spectrum/mod.rs:601-602)."""
    O = oracle
    xyz = O.to_xyz(O.from_rgb((1.0, 1.0, 1.0)))
    out = (C.c_float * 3)()
    O.lib().ref_xyz_to_rgb(xyz.ctypes.data_as(C.POINTER(C.c_float)), out)
    rows = np.array([3.2404790 - 1.537150 - 0.498535, -0.969256 + 1.875991 + 0.041556, 0.0556480 - 0.204043 + 1.057311])
    assert np.allclose(list(out), 1.0615 * rows, atol=0.01)
    one = np.ones(60, dtype=np.float32)
    assert abs(O.to_xyz(one)[1] - 1.0) < 0.01


@pytest.mark.parametrize("rgb,expect", [
    ((0.0, 0.0, 0.0), (0, 0, 0)), ((1.0, 1.0, 1.0), (255, 255, 255)), ((2.0, -1.0, 0.5), (255, 0, 186)),
    ((float("nan"), 0.25, 1e-9), (0, 136, 0)),
])
def test_tonemap(oracle, rgb, expect):
    """src/integrators/integrator.rs:80-90: clamp, ^(1/2.2), (256*v) as u8 (saturating, NaN -> 0)."""
    out = (C.c_uint8 * 3)()
    oracle.lib().ref_tonemap((C.c_float * 3)(*rgb), out)
    assert tuple(out) == expect


def test_image_orientation_and_determinism(oracle):
    """src/camera.rs:26-29 + src/integrators/integrator.rs:110-112: y = 0 is the top row. An emitter above
    the optical axis lights the upper half. Same seed -> same image; another seed -> another image."""
    O = oracle
    b = SceneBuilder(O.from_rgb)
    b.add_sphere((5, 0, 3), 1.5, material=b.material("Emitter", (1, 1, 1)))
    s = O.Scene(*b.arrays())
    cam = O.camera((0, 0, 0), 1.0, 90.0, (0, 0, 0), 64, 64)
    a, _, _, st = s.render(cam, O.INT_RANDOM_WALK, 3, 4, seed=1)
    assert a[:32].sum() > 0 and a[40:].sum() == 0
    assert st.paths == 64 * 64 * 4 and st.segments == st.paths
    a2, *_ = s.render(cam, O.INT_RANDOM_WALK, 3, 4, seed=1, threads=1)
    a3, *_ = s.render(cam, O.INT_RANDOM_WALK, 3, 4, seed=2)
    assert (a == a2).all() and (a != a3).any()


def test_pixel_jitter_covers_the_pixel(oracle):
    """src/integrators/integrator.rs:130-132: x + U - 0.5 with U in [0,1)."""
    O = oracle
    out = O.vec3(0, 0, 0)
    u = []
    for s in range(512):
        O.lib().ref_uniform_pair(11, 5, s, 0, 0, out)
        u.append((out[0], out[1]))
    u = np.array(u)
    assert u.min() >= 0 and u.max() < 1 and (np.abs(u.mean(axis=0) - 0.5) < 0.05).all()


def test_optional_cpu_bvh_returns_the_linear_scan(oracle):
    """ref_scene_enable_bvh is not part of the restatement (the reference scans every triangle); it only serves the
    accelerated CPU baseline. It must not change a single hit or path under either rule."""
    import numpy as np
    import _scenes as S
    O = oracle
    for rule in (O.HIT_FIRST_INDEX, O.HIT_NEAREST):
        plain, cam, _ = S.oracle_scene(S.multi_mesh_scene, hit_rule=rule, width=96, height=54)
        fast, _, _ = S.oracle_scene(S.multi_mesh_scene, hit_rule=rule, width=96, height=54)
        fast.enable_bvh()
        rays = np.concatenate([S.primary_rays(cam), S.random_rays(6000, 3, center=(0.5, 0.0, 1.0), radius=5.0)])
        assert plain.intersect_batch(rays).tobytes() == fast.intersect_batch(rays).tobytes()
        a = plain.render(cam, O.INT_RANDOM_WALK, 8, 2, seed=3)
        b = fast.render(cam, O.INT_RANDOM_WALK, 8, 2, seed=3)
        assert (a[0] == b[0]).all() and a[3].segments == b[3].segments
