"""ctypes binding of the CPU oracle (oracle/librefcpu.so). Test infrastructure only."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB_PATH = ROOT / "oracle" / "librefcpu.so"

NBINS = 60
MISS = 0xFFFFFFFF
OBJ_SPHERE, OBJ_TRIANGLE, OBJ_MODEL = 0, 1, 2
MAT_EMITTER, MAT_LAMBERTIAN, MAT_SPECULAR, MAT_GLOSSY, MAT_DIELECTRIC = 0, 1, 2, 3, 4
HAS_AMBIENT, HAS_EMITTED, HAS_DIFFUSED, HAS_SPECULAR = 1, 2, 4, 8
HIT_FIRST_INDEX, HIT_NEAREST = 0, 1
INT_RANDOM_WALK, INT_ALBEDO, INT_NORMAL = 0, 1, 2
SPEC_REFLECTANCE, SPEC_ILLUMINANT = 0, 1

# numpy record layouts shared by the oracle and the product library (identical C structs)
RAY_DT = np.dtype([("o", "<f8", 3), ("d", "<f8", 3)])
HIT_DT = np.dtype([("obj", "<u4"), ("prim", "<u4"), ("t", "<f8"), ("p", "<f8", 3), ("n", "<f8", 3)])
OBJECT_DT = np.dtype([("kind", "<u4"), ("first", "<u4"), ("count", "<u4"), ("material", "<u4")])
SPHERE_DT = np.dtype([("l2w", "<f8", 12), ("radius", "<f8")])
TRIANGLE_DT = np.dtype([("v", "<f8", 9)])
MATERIAL_DT = np.dtype([("type", "<u4"), ("present", "<u4"), ("ambient", "<f4", NBINS), ("emitted", "<f4", NBINS),
                        ("diffused", "<f4", NBINS), ("specular", "<f4", NBINS),
                        ("specular_coefficient", "<f4"), ("optical_density", "<f4")])


class Affine(C.Structure):
    _fields_ = [("m", C.c_double * 12)]


class Ray(C.Structure):
    _fields_ = [("o", C.c_double * 3), ("d", C.c_double * 3)]


class Hit(C.Structure):
    _fields_ = [("obj", C.c_uint32), ("prim", C.c_uint32), ("t", C.c_double), ("p", C.c_double * 3), ("n", C.c_double * 3)]


class Sphere(C.Structure):
    _fields_ = [("l2w", C.c_double * 12), ("radius", C.c_double)]


class Triangle(C.Structure):
    _fields_ = [("v", C.c_double * 9)]


class SceneDesc(C.Structure):
    _fields_ = [("objects", C.c_void_p), ("n_objects", C.c_uint32),
                ("spheres", C.c_void_p), ("n_spheres", C.c_uint32),
                ("triangles", C.c_void_p), ("n_triangles", C.c_uint32),
                ("materials", C.c_void_p), ("n_materials", C.c_uint32),
                ("hit_rule", C.c_uint32), ("reserved", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("focus", C.c_double * 3), ("p1", C.c_double * 3), ("p2", C.c_double * 3), ("p3", C.c_double * 3),
                ("p4", C.c_double * 3), ("width", C.c_uint32), ("height", C.c_uint32)]


class RenderParams(C.Structure):
    _fields_ = [("integrator", C.c_uint32), ("max_depth", C.c_uint32), ("spp", C.c_uint32), ("tile_size", C.c_uint32),
                ("seed", C.c_uint64), ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32),
                ("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32),
                ("tile_rank", C.c_uint32), ("tile_world", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("segments", C.c_uint64), ("paths", C.c_uint64), ("seconds", C.c_double), ("threads", C.c_uint32),
                ("pad", C.c_uint32)]


def build():
    """Compile the oracle with its own Makefile (gcc only)."""
    subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        build()
    L = C.CDLL(str(LIB_PATH))
    d3 = C.POINTER(C.c_double)
    f60 = C.POINTER(C.c_float)
    L.ref_spectrum_init.restype = None
    L.ref_average_spectrum_sample.restype = C.c_float
    L.ref_average_spectrum_sample.argtypes = [f60, f60, C.c_int, C.c_float, C.c_float]
    L.ref_from_rgb.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, f60]
    L.ref_to_xyz.argtypes = [f60, f60]
    L.ref_xyz_to_rgb.argtypes = [f60, f60]
    L.ref_tonemap.argtypes = [f60, C.POINTER(C.c_uint8)]
    L.ref_spectrum_tables.argtypes = [f60, f60, f60]
    L.ref_smits_table.argtypes = [C.c_int, C.c_int, f60]
    L.ref_affine_translation.argtypes = [d3, C.POINTER(Affine)]
    L.ref_affine_scaling.argtypes = [d3, C.POINTER(Affine)]
    L.ref_affine_rotation.argtypes = [d3, C.c_double, C.POINTER(Affine)]
    L.ref_affine_mul.argtypes = [C.POINTER(Affine)] * 3
    L.ref_affine_inverse.argtypes = [C.POINTER(Affine)] * 2
    L.ref_affine_point.argtypes = [C.POINTER(Affine), d3, d3]
    L.ref_affine_vector.argtypes = [C.POINTER(Affine), d3, d3]
    L.ref_ray_new.argtypes = [d3, d3, C.POINTER(Ray)]
    L.ref_ray_at.argtypes = [C.POINTER(Ray), C.c_double, d3]
    L.ref_affine_ray.argtypes = [C.POINTER(Affine), C.POINTER(Ray), C.POINTER(Ray)]
    L.ref_intersect_ray_and_triangle.argtypes = [d3, C.POINTER(Ray), d3]
    L.ref_aabb_intersect_ray.argtypes = [d3, d3, C.POINTER(Ray), d3]
    for f in (L.ref_smaller_of_three, L.ref_bigger_of_three):
        f.argtypes = [C.c_double] * 3
        f.restype = C.c_double
    L.ref_aabb_new.argtypes = [d3, d3, d3, d3]
    L.ref_aabb_new.restype = None
    L.ref_aabb_union.argtypes = [d3, d3, d3, d3, d3, d3]
    L.ref_aabb_union.restype = None
    L.ref_random_versor_cos_density.argtypes = [d3, C.c_double, C.c_double, d3]
    L.ref_phong_lobe_direction.argtypes = [d3, C.c_double, C.c_double, C.c_double, d3]
    L.ref_dielectric_direction.restype = C.c_int
    L.ref_dielectric_direction.argtypes = [d3, d3, C.c_int, C.c_double, C.c_double, d3]
    L.ref_sphere_intersect.argtypes = [C.POINTER(Sphere), C.POINTER(Ray), C.POINTER(Hit)]
    L.ref_triangle_intersect.argtypes = [C.POINTER(Triangle), C.POINTER(Affine), C.POINTER(Ray), C.POINTER(Hit)]
    L.ref_sphere_bounds.argtypes = [C.POINTER(Sphere), d3, d3]
    L.ref_camera_new.argtypes = [d3, C.c_double, C.c_double, d3, C.c_uint32, C.c_uint32, C.POINTER(Camera)]
    L.ref_camera_get_ray.argtypes = [C.POINTER(Camera), C.c_double, C.c_double, C.POINTER(Ray)]
    L.ref_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
    L.ref_uniform_pair.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, d3]
    L.ref_scene_create.restype = C.c_void_p
    L.ref_scene_create.argtypes = [C.POINTER(SceneDesc)]
    L.ref_scene_destroy.argtypes = [C.c_void_p]
    L.ref_scene_enable_bvh.argtypes = [C.c_void_p, C.c_int]
    L.ref_scene_model_bounds.argtypes = [C.c_void_p, C.c_uint32, d3, d3]
    L.ref_scene_intersect.argtypes = [C.c_void_p, C.POINTER(Ray), C.POINTER(Hit)]
    L.ref_intersect_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    L.ref_trace_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32,
                                  C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    L.ref_evaluate_pixel_sample.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_uint32,
                                            C.c_uint32, C.c_uint32, f60, C.POINTER(C.c_uint64)]
    L.ref_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.POINTER(Stats), C.c_int]
    L.ref_spectrum_init()
    _lib = L
    return L


def vec3(x, y, z):
    return (C.c_double * 3)(x, y, z)


def make_ray(o, d):
    r = Ray()
    lib().ref_ray_new(vec3(*o), vec3(*d), C.byref(r))
    return r


def from_rgb(rgb, kind=SPEC_REFLECTANCE):
    out = (C.c_float * NBINS)()
    lib().ref_from_rgb(C.c_float(rgb[0]), C.c_float(rgb[1]), C.c_float(rgb[2]), kind, out)
    return np.frombuffer(out, dtype=np.float32).copy()


def to_xyz(spec):
    s = np.ascontiguousarray(spec, dtype=np.float32)
    out = (C.c_float * 3)()
    lib().ref_to_xyz(s.ctypes.data_as(C.POINTER(C.c_float)), out)
    return np.array(list(out), dtype=np.float32)


def camera(focus, focal_distance, fov, rotation, width, height):
    c = Camera()
    lib().ref_camera_new(vec3(*focus), focal_distance, fov, vec3(*rotation), width, height, C.byref(c))
    return c


class Scene:
    """Owns an oracle scene built from numpy record arrays (see *_DT above)."""

    def __init__(self, objects, spheres, triangles, materials, hit_rule=HIT_FIRST_INDEX):
        self.objects = np.ascontiguousarray(objects, dtype=OBJECT_DT)
        self.spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DT)
        self.triangles = np.ascontiguousarray(triangles, dtype=TRIANGLE_DT)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL_DT)
        d = SceneDesc(self.objects.ctypes.data, len(self.objects), self.spheres.ctypes.data, len(self.spheres),
                      self.triangles.ctypes.data, len(self.triangles), self.materials.ctypes.data, len(self.materials),
                      hit_rule, 0)
        self.handle = lib().ref_scene_create(C.byref(d))

    def __del__(self):
        if getattr(self, "handle", None):
            lib().ref_scene_destroy(self.handle)
            self.handle = None

    def enable_bvh(self, on=True):
        """Not the reference's algorithm: an accelerated CPU baseline with the linear scan's results (timing only)."""
        lib().ref_scene_enable_bvh(self.handle, 1 if on else 0)

    def intersect_batch(self, rays, threads=None):
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        out = np.zeros(len(rays), dtype=HIT_DT)
        lib().ref_intersect_batch(self.handle, rays.ctypes.data, len(rays), out.ctypes.data, threads or (os.cpu_count() or 1))
        return out

    def trace_batch(self, rays, integrator, max_depth, seed, pixel_base=0, sample_index=0, threads=None):
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        out = np.zeros((len(rays), NBINS), dtype=np.float32)
        seg = C.c_uint64(0)
        lib().ref_trace_batch(self.handle, rays.ctypes.data, len(rays), integrator, max_depth, seed, pixel_base,
                              sample_index, out.ctypes.data, C.byref(seg), threads or (os.cpu_count() or 1))
        return out, seg.value

    def render(self, cam, integrator, max_depth, spp, seed=0, sample_range=None, window=None, threads=None,
               want_lin=True, tile_rank=0, tile_world=0, xyz_sum=None):
        p = RenderParams(integrator, max_depth, spp, 16, seed, 0, 0, 0, 0, 0, 0, tile_rank, tile_world)
        if sample_range:
            p.sample_begin, p.sample_end = sample_range
        if window:
            p.x0, p.y0, p.x1, p.y1 = window
        w, h = cam.width, cam.height
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        xyz_ref = np.zeros((h, w, 3), dtype=np.float32)
        xyz_lin = np.zeros((h, w, 3), dtype=np.float32) if want_lin else None
        st = Stats()
        lib().ref_render(self.handle, C.byref(cam), C.byref(p), rgb8.ctypes.data, xyz_ref.ctypes.data,
                         xyz_lin.ctypes.data if want_lin else None, xyz_sum.ctypes.data if xyz_sum is not None else None,
                         C.byref(st), threads or (os.cpu_count() or 1))
        return rgb8, xyz_ref, xyz_lin, st
