"""ctypes binding of librtmi.so (the C ABI declared in include/rtmi.h).

There is no CPU fallback: if the shared library is missing this module raises, and if no GPU is
visible rtmi_ctx_create reports RTMI_NO_DEVICE.
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

import os

PKG = Path(__file__).resolve().parent
# RTMI_LIB selects another build of the same library (A/B experiments); the default is the in-tree one.
LIB_PATH = Path(os.environ["RTMI_LIB"]).resolve() if os.environ.get("RTMI_LIB") else PKG / "librtmi.so"

NBINS = 60
MISS = 0xFFFFFFFF
MAX_DEPTH = 32
OK, BAD_ARG, NO_DEVICE, OOM, HIP_ERROR, UNSUPPORTED, RCCL_ERROR, INTERNAL = 0, -1, -2, -3, -4, -5, -6, -7
OBJ_SPHERE, OBJ_TRIANGLE, OBJ_MODEL = 0, 1, 2
MAT_EMITTER, MAT_LAMBERTIAN, MAT_SPECULAR, MAT_GLOSSY, MAT_DIELECTRIC = 0, 1, 2, 3, 4
HAS_AMBIENT, HAS_EMITTED, HAS_DIFFUSED, HAS_SPECULAR = 1, 2, 4, 8
HIT_FIRST_INDEX, HIT_NEAREST = 0, 1
INT_RANDOM_WALK, INT_ALBEDO, INT_NORMAL = 0, 1, 2
SPEC_REFLECTANCE, SPEC_ILLUMINANT = 0, 1
FLAG_TIME_KERNELS, FLAG_COUNT_TRAVERSAL = 1, 2
PARTITION_SAMPLES, PARTITION_TILES = 0, 1
SCENE_GPU_BVH = 1

# numpy record layouts of the plain C structs
RAY_DT = np.dtype([("o", "<f8", 3), ("d", "<f8", 3)])
HIT_DT = np.dtype([("obj", "<u4"), ("prim", "<u4"), ("t", "<f8"), ("p", "<f8", 3), ("n", "<f8", 3)])
OBJECT_DT = np.dtype([("kind", "<u4"), ("first", "<u4"), ("count", "<u4"), ("material", "<u4")])
SPHERE_DT = np.dtype([("l2w", "<f8", 12), ("radius", "<f8")])
TRIANGLE_DT = np.dtype([("v", "<f8", 9)])
MATERIAL_DT = np.dtype([("type", "<u4"), ("present", "<u4"), ("ambient", "<f4", NBINS), ("emitted", "<f4", NBINS),
                        ("diffused", "<f4", NBINS), ("specular", "<f4", NBINS),
                        ("specular_coefficient", "<f4"), ("optical_density", "<f4")])


class SceneDesc(C.Structure):
    _fields_ = [("objects", C.c_void_p), ("n_objects", C.c_uint32),
                ("spheres", C.c_void_p), ("n_spheres", C.c_uint32),
                ("triangles", C.c_void_p), ("n_triangles", C.c_uint32),
                ("materials", C.c_void_p), ("n_materials", C.c_uint32),
                ("hit_rule", C.c_uint32), ("flags", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("focus", C.c_double * 3), ("p1", C.c_double * 3), ("p2", C.c_double * 3), ("p3", C.c_double * 3),
                ("p4", C.c_double * 3), ("width", C.c_uint32), ("height", C.c_uint32)]


class Ray(C.Structure):
    _fields_ = [("o", C.c_double * 3), ("d", C.c_double * 3)]


class RenderParams(C.Structure):
    _fields_ = [("integrator", C.c_uint32), ("max_depth", C.c_uint32), ("spp", C.c_uint32), ("flags", C.c_uint32),
                ("seed", C.c_uint64), ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32),
                ("tile_rank", C.c_uint32), ("tile_world", C.c_uint32), ("tile_size", C.c_uint32), ("reserved", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("segments", C.c_uint64), ("paths", C.c_uint64), ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64),
                ("ms_total", C.c_double), ("ms_kernels", C.c_double), ("launches", C.c_uint32), ("batches", C.c_uint32),
                ("mesh_rays", C.c_uint64), ("ms_traverse", C.c_double), ("traverse_launches", C.c_uint32), ("reserved", C.c_uint32)]


class MultiInfo(C.Structure):
    _fields_ = [("n_ranks", C.c_uint32), ("uses_rccl", C.c_uint32), ("rccl_library", C.c_char * 240)]


class SceneInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("max_depth", C.c_uint32), ("n_mesh_triangles", C.c_uint32), ("n_spectra", C.c_uint32),
                ("node_bytes", C.c_uint64), ("triangle_bytes", C.c_uint64), ("bvh_build_ms", C.c_double), ("bvh_device_ms", C.c_double), ("upload_ms", C.c_double)]


# every symbol include/rtmi.h declares: name -> (restype, argtypes)
_f = C.POINTER(C.c_float)
_d = C.POINTER(C.c_double)
SYMBOLS = {
    "rtmi_ctx_create": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rtmi_ctx_destroy": (None, [C.c_void_p]),
    "rtmi_last_error": (C.c_char_p, []),
    "rtmi_version": (C.c_char_p, []),
    "rtmi_spectrum_from_rgb": (C.c_int, [C.c_float, C.c_float, C.c_float, C.c_int, _f]),
    "rtmi_spectrum_to_xyz": (C.c_int, [_f, _f]),
    "rtmi_xyz_to_rgb": (C.c_int, [_f, _f]),
    "rtmi_spectrum_cie_tables": (C.c_int, [_f, _f, _f]),
    "rtmi_camera_new": (C.c_int, [_d, C.c_double, C.c_double, _d, C.c_uint32, C.c_uint32, C.POINTER(Camera)]),
    "rtmi_camera_get_ray": (C.c_int, [C.POINTER(Camera), C.c_double, C.c_double, C.POINTER(Ray)]),
    "rtmi_scene_create": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]),
    "rtmi_scene_destroy": (None, [C.c_void_p]),
    "rtmi_scene_get_info": (C.c_int, [C.c_void_p, C.POINTER(SceneInfo)]),
    "rtmi_intersect_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Stats)]),
    "rtmi_intersect_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "rtmi_trace_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32,
                                   C.c_uint32, C.c_void_p, C.POINTER(Stats)]),
    "rtmi_render": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "rtmi_render_accumulate": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.c_void_p,
                                         C.POINTER(Stats)]),
    "rtmi_render_accumulate_noalloc": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.c_void_p]),
    "rtmi_render_collect_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "rtmi_multi_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_uint64, C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]),
    "rtmi_multi_destroy": (None, [C.c_void_p]),
    "rtmi_multi_info": (C.c_int, [C.c_void_p, C.POINTER(MultiInfo)]),
    "rtmi_multi_render": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.POINTER(Stats), C.POINTER(Stats)]),
    "rtmi_multi_render_pass": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "rtmi_tonemap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
}


class RtmiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rtmi error %d: %s" % (code, message))
        self.code = code


def build(verbose=False):
    """Compile librtmi.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", str(PKG / "csrc")], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout, r.stderr)
    if r.returncode:
        raise RuntimeError("building librtmi.so failed")


_lib = None
_hip_runtime = None


def _preload_hip_runtime():
    """librtmi.so leaves the HIP runtime symbols undefined so that it shares the runtime the process
    already uses. PyTorch bundles its own libamdhip64.so; when torch is importable that copy is
    made global, otherwise the system ROCm one is."""
    global _hip_runtime
    if _hip_runtime is not None:
        return
    candidates = []
    try:
        import torch
        candidates.append(Path(torch.__file__).resolve().parent / "lib" / "libamdhip64.so")
    except Exception:
        pass
    candidates += [Path("/opt/rocm/lib/libamdhip64.so"), Path("libamdhip64.so")]
    for cand in candidates:
        try:
            _hip_runtime = C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
            return
        except OSError:
            continue
    raise RuntimeError("no HIP runtime (libamdhip64.so) could be loaded")


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError("%s is missing: build it with raytracing_amd.build() (there is no CPU fallback)" % LIB_PATH)
        _preload_hip_runtime()
        L = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(code):
    if code != OK:
        raise RtmiError(code, lib().rtmi_last_error().decode())
    return code
