"""Pins the CPU oracle against every known-answer test the reference holds for the hot path.

Each test names the reference test it replays (file:line in the reference repository). The
tolerance is the reference's own: assert_eq_float! is |a-b| < 1e-10 (src/auxiliar.rs:33-47).
"""
import ctypes as C
import math

import numpy as np
import pytest

TOL = 1e-10


def close(a, b):
    return all(abs(x - y) < TOL for x, y in zip(a, b))


def affine(o, kind, v, angle=None):
    a = o.Affine()
    if kind == "t":
        o.lib().ref_affine_translation(o.vec3(*v), C.byref(a))
    elif kind == "s":
        o.lib().ref_affine_scaling(o.vec3(*v), C.byref(a))
    else:
        o.lib().ref_affine_rotation(o.vec3(*v), angle, C.byref(a))
    return a


def mul(o, a, b):
    r = o.Affine()
    o.lib().ref_affine_mul(C.byref(a), C.byref(b), C.byref(r))
    return r


def apply_point(o, a, p):
    out = o.vec3(0, 0, 0)
    o.lib().ref_affine_point(C.byref(a), o.vec3(*p), out)
    return list(out)


def apply_vector(o, a, p):
    out = o.vec3(0, 0, 0)
    o.lib().ref_affine_vector(C.byref(a), o.vec3(*p), out)
    return list(out)


# ---- src/camera.rs:94-164

def test_camera_without_rotation(oracle):  # crear_cámara_sin_rotación, src/camera.rs:94
    c = oracle.camera((0, 0, 0), 1.0, 90.0, (0, 0, 0), 500, 1000)
    assert close(c.focus, (0, 0, 0))
    assert close(c.p1, (1.0, -1.0, 2.0))
    assert close(c.p4, (1.0, 1.0, -2.0))
    assert (c.width, c.height) == (500, 1000)


def test_camera_with_rotation(oracle):  # crear_cámara_con_rotación, src/camera.rs:118
    c = oracle.camera((0, 0, 0), 1.0, 90.0, (math.pi / 2, 0, 0), 500, 1000)
    assert close(c.p1, (1.0, -2.0, -1.0))
    assert close(c.p4, (1.0, 2.0, 1.0))


def test_camera_get_ray(oracle):  # lanzar_rayo, src/camera.rs:142
    c = oracle.camera((0, 0, 0), 1.0, 90.0, (0, 0, 0), 100, 100)
    r = oracle.Ray()
    oracle.lib().ref_camera_get_ray(C.byref(c), 0.0, 0.0, C.byref(r))
    k = 1 / math.sqrt(3)
    assert close(r.o, (0, 0, 0)) and close(r.d, (k, -k, k))
    oracle.lib().ref_camera_get_ray(C.byref(c), 50.0, 50.0, C.byref(r))
    assert close(r.d, (1, 0, 0))


# ---- src/geometry/mod.rs:123-152

TRI = (1.0, -1.0, 0.0, 1.0, 1.0, 0.0, 1.0, 0.0, 1.0)


def test_moller_trumbore_hit(oracle):  # triángulo_interseca_rayo, src/geometry/mod.rs:123
    r = oracle.make_ray((0, 0, 0), (1.0, 0.0, 0.5))
    tuv = oracle.vec3(0, 0, 0)
    assert oracle.lib().ref_intersect_ray_and_triangle((C.c_double * 9)(*TRI), C.byref(r), tuv) == 1


def test_moller_trumbore_miss(oracle):  # triángulo_no_interseca_rayo, src/geometry/mod.rs:139
    r = oracle.make_ray((0, 0, 0), (1.0, 3.0, 3.0))
    tuv = oracle.vec3(0, 0, 0)
    assert oracle.lib().ref_intersect_ray_and_triangle((C.c_double * 9)(*TRI), C.byref(r), tuv) == 0


# ---- src/geometry/ray.rs:68-127

def test_ray_at(oracle):  # evaluar_rayo, src/geometry/ray.rs:68
    r = oracle.make_ray((1, 1, 2), (2, 2, 1))
    out = oracle.vec3(0, 0, 0)
    assert oracle.lib().ref_ray_at(C.byref(r), 3.0, out) == 1
    assert close(out, (3, 3, 3))
    assert oracle.lib().ref_ray_at(C.byref(r), -1.0, out) == 0


def test_ray_translation(oracle):  # translation, src/geometry/ray.rs:85
    r = oracle.make_ray((1, 0, 0), (0, 0, 1))
    q = oracle.Ray()
    oracle.lib().ref_affine_ray(C.byref(affine(oracle, "t", (1, 0, 0))), C.byref(r), C.byref(q))
    assert close(q.o, (2, 0, 0)) and close(q.d, (0, 0, 1))


def test_ray_rotation(oracle):  # rotation, src/geometry/ray.rs:100
    r = oracle.make_ray((1, 0, 0), (0, 0, 1))
    q = oracle.Ray()
    oracle.lib().ref_affine_ray(C.byref(affine(oracle, "r", (0, 1, 0), math.pi / 2)), C.byref(r), C.byref(q))
    assert close(q.o, (0, 0, -1)) and close(q.d, (1, 0, 0))


def test_ray_scaling(oracle):  # scaling, src/geometry/ray.rs:115
    r = oracle.make_ray((1, 0, 0), (0, 1, 1))
    q = oracle.Ray()
    oracle.lib().ref_affine_ray(C.byref(affine(oracle, "s", (2, 0.5, 3))), C.byref(r), C.byref(q))
    n = math.sqrt(0.25 + 9)
    assert close(q.o, (2, 0, 0)) and close(q.d, (0, 0.5 / n, 3 / n))


# ---- src/geometry/common.rs:32-96

def test_transforms(oracle):  # compose_transforms / transform_point / transform_vector
    t = affine(oracle, "t", (1, 2, 3))
    r = affine(oracle, "r", (0, 0, 1), math.pi / 2)
    s = affine(oracle, "s", (2, 3, 4))
    p = (1, 2, 3)
    rs = mul(oracle, r, s)
    trs = mul(oracle, t, rs)
    assert close(apply_point(oracle, r, apply_point(oracle, s, p)), apply_point(oracle, rs, p))
    assert close(apply_point(oracle, rs, p), (-6, 2, 12))
    assert close(apply_point(oracle, trs, p), (-5, 4, 15))
    assert close(apply_point(oracle, s, p), (2, 6, 12))
    assert close(apply_point(oracle, r, p), (-2, 1, 3))
    assert close(apply_point(oracle, t, p), (2, 4, 6))
    assert close(apply_vector(oracle, s, p), (2, 6, 12))
    assert close(apply_vector(oracle, r, p), (-2, 1, 3))
    assert close(apply_vector(oracle, t, p), (1, 2, 3))
    assert close(apply_vector(oracle, trs, p), (-6, 2, 12))


def test_affine_inverse_roundtrip(oracle):
    trs = mul(oracle, affine(oracle, "t", (1, 2, 3)), mul(oracle, affine(oracle, "r", (0, 0, 1), 0.7), affine(oracle, "s", (2, 3, 4))))
    inv = oracle.Affine()
    assert oracle.lib().ref_affine_inverse(C.byref(trs), C.byref(inv)) == 1
    p = (0.3, -1.2, 2.5)
    assert close(apply_point(oracle, inv, apply_point(oracle, trs, p)), p)


# ---- src/geometry/aabb.rs:256-318

def aabb_hit(o, bmin, bmax, origin, direction):
    r = o.make_ray(origin, direction)
    t = o.vec3(0, 0, 0)
    return o.lib().ref_aabb_intersect_ray(o.vec3(*bmin), o.vec3(*bmax), C.byref(r), t) == 1


def test_aabb(oracle):
    assert aabb_hit(oracle, (1, 1, 1), (2, 2, 2), (0, 0, 0), (1.5, 1.5, 1.5))      # caja_interseca_rayo :256
    assert not aabb_hit(oracle, (1, 1, 1), (2, 2, 2), (0, 0, 0), (1, 0, 0))        # caja_no_interseca_rayo :269
    assert aabb_hit(oracle, (0, 0, 0), (2, 2, 2), (1, 1, 1), (1, 0, 0))            # rayo_adentro_de_caja :282


def test_smaller_and_bigger_of_three(oracle):  # src/auxiliar.rs:54,63 (the corners of a triangle's box, shapes/triangle.rs:44-54)
    a, b, c = 1.0, 2.0, 3.0
    lo, hi = oracle.lib().ref_smaller_of_three, oracle.lib().ref_bigger_of_three
    assert abs(lo(a, b, c) - a) < TOL and abs(lo(c, b, a) - a) < TOL and abs(lo(b, a, c) - a) < TOL
    assert abs(hi(a, b, c) - c) < TOL and abs(hi(c, b, a) - c) < TOL and abs(hi(b, c, a) - c) < TOL


def aabb_box(o, p1, p2):
    lo, hi = o.vec3(0, 0, 0), o.vec3(0, 0, 0)
    o.lib().ref_aabb_new(o.vec3(*p1), o.vec3(*p2), lo, hi)
    return lo, hi


def test_aabb_union(oracle):
    """unir_cajas :295 and ampliar_caja :308 (resize_box is the union in place, aabb.rs:77-79): the fold ModelObj::new runs
    over its triangles' boxes to get the box every mesh ray is tested against first (shapes/model_obj.rs:62-66,78)."""
    c1, c2 = aabb_box(oracle, (1, 1, 1), (2, 2, 2)), aabb_box(oracle, (0, 0, 0), (1, 1, 1))
    lo, hi = oracle.vec3(0, 0, 0), oracle.vec3(0, 0, 0)
    oracle.lib().ref_aabb_union(c1[0], c1[1], c2[0], c2[1], lo, hi)
    assert close(list(lo), (0, 0, 0)) and close(list(hi), (2, 2, 2))
    oracle.lib().ref_aabb_union(c1[0], c1[1], c2[0], c2[1], c1[0], c1[1])          # in place, as resize_box
    assert close(list(c1[0]), (0, 0, 0)) and close(list(c1[1]), (2, 2, 2))
    # AABB::new orders its corners (aabb.rs:26-39)
    lo, hi = aabb_box(oracle, (2, 0, 5), (1, 3, 4))
    assert list(lo) == [1, 0, 4] and list(hi) == [2, 3, 5]


# ---- src/shapes/triangle.rs:110-133 (the :136 test is wrong upstream: 90.0 is radians there)

def test_triangle_intersects_ray(oracle):
    tri = oracle.Triangle((C.c_double * 9)(0, 0, 0, 1, 0, 0, 0, 1, 0))
    ident = oracle.Affine()
    oracle.lib().ref_affine_identity(C.byref(ident))
    r = oracle.make_ray((0, 0, -1), (0, 0, 1))
    h = oracle.Hit()
    assert oracle.lib().ref_triangle_intersect(C.byref(tri), C.byref(ident), C.byref(r), C.byref(h)) == 1
    assert abs(h.t - 1.0) < TOL and close(h.p, (0, 0, 0)) and close(h.n, (0, 0, 1))


# ---- src/shapes/sphere.rs:117-254

def sphere(o, a, radius=1.0):
    s = o.Sphere()
    for i in range(12):
        s.l2w[i] = a.m[i]
    s.radius = radius
    return s


def identity(o):
    a = o.Affine()
    o.lib().ref_affine_identity(C.byref(a))
    return a


def test_sphere_intersects_ray(oracle):  # :117
    s = sphere(oracle, identity(oracle))
    h = oracle.Hit()
    r = oracle.make_ray((0, 0, -5), (0, 0, 1))
    assert oracle.lib().ref_sphere_intersect(C.byref(s), C.byref(r), C.byref(h)) == 1
    assert abs(h.t - 4.0) < TOL and close(h.p, (0, 0, -1)) and close(h.n, (0, 0, -1))


def test_translated_sphere(oracle):  # :139
    s = sphere(oracle, affine(oracle, "t", (2, 0, 0)))
    h = oracle.Hit()
    r = oracle.make_ray((0, 0, 0), (1, 0, 0))
    assert oracle.lib().ref_sphere_intersect(C.byref(s), C.byref(r), C.byref(h)) == 1
    assert abs(h.t - 1.0) < TOL and close(h.p, (1, 0, 0)) and close(h.n, (-1, 0, 0))


def test_sphere_bounds(oracle):  # aabb_in_sphere_* :163,:174
    lo, hi = oracle.vec3(0, 0, 0), oracle.vec3(0, 0, 0)
    oracle.lib().ref_sphere_bounds(C.byref(sphere(oracle, identity(oracle))), lo, hi)
    assert list(lo) == [-1, -1, -1] and list(hi) == [1, 1, 1]
    oracle.lib().ref_sphere_bounds(C.byref(sphere(oracle, affine(oracle, "t", (0, 0, 1)))), lo, hi)
    assert list(lo) == [-1, -1, 0] and list(hi) == [1, 1, 2]


def test_uniformly_scaled_sphere(oracle):  # :188 (t is in local units, so the reference does not assert it)
    s = sphere(oracle, affine(oracle, "s", (0.5, 0.5, 0.5)))
    h = oracle.Hit()
    r = oracle.make_ray((0, 0, -1), (0, 0, 1))
    assert oracle.lib().ref_sphere_intersect(C.byref(s), C.byref(r), C.byref(h)) == 1
    assert close(h.p, (0, 0, -0.5)) and close(h.n, (0, 0, -1))
    assert abs(h.t - 1.0) < TOL  # local-space t: the unit sphere is 1 away along the local ray


def test_non_uniformly_scaled_sphere(oracle):  # :214
    scale = affine(oracle, "s", (1.0, 0.5, 1.0))
    plain = sphere(oracle, identity(oracle))
    aux = oracle.make_ray((0, -2, 1), (0, 2, -1))
    h0 = oracle.Hit()
    assert oracle.lib().ref_sphere_intersect(C.byref(plain), C.byref(aux), C.byref(h0)) == 1
    expected_point = apply_point(oracle, scale, list(h0.p))
    n = [h0.n[0] / 1.0, h0.n[1] / 0.5, h0.n[2] / 1.0]  # (S^-1)^T n
    ln = math.sqrt(sum(x * x for x in n))
    expected_normal = [x / ln for x in n]
    s = sphere(oracle, scale)
    r = oracle.make_ray((0, -1, 1), (0, 1, -1))
    h = oracle.Hit()
    assert oracle.lib().ref_sphere_intersect(C.byref(s), C.byref(r), C.byref(h)) == 1
    assert close(h.p, expected_point) and close(h.n, expected_normal)


# ---- Philox4x32-10 known answers (Random123 kat_vectors)

@pytest.mark.parametrize("ctr,key,expect", [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
])
def test_philox_kat(oracle, ctr, key, expect):
    out = (C.c_uint32 * 4)()
    oracle.lib().ref_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    assert tuple(out) == expect


def test_uniform_pair_range(oracle):
    out = oracle.vec3(0, 0, 0)
    vals = []
    for i in range(2000):
        oracle.lib().ref_uniform_pair(7, i, 3, 1, 0, out)
        vals += [out[0], out[1]]
    v = np.array(vals)
    assert v.min() >= 0.0 and v.max() < 1.0 and abs(v.mean() - 0.5) < 0.02
