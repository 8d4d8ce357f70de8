"""Thin Python handles over the C ABI: Context, Scene, camera and spectrum helpers.

Plumbing only -- every call goes straight into librtmi.so. Arrays are numpy record arrays with
the layouts in _lib.py; device buffers are passed as integer addresses (e.g. tensor.data_ptr()).
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib as L


def spectrum_from_rgb(rgb, kind=L.SPEC_REFLECTANCE):
    """SampledSpectrum::from_RGB (src/spectrum/mod.rs:525-581)."""
    out = (C.c_float * L.NBINS)()
    L.check(L.lib().rtmi_spectrum_from_rgb(C.c_float(rgb[0]), C.c_float(rgb[1]), C.c_float(rgb[2]), int(kind), out))
    return np.frombuffer(out, dtype=np.float32).copy()


def spectrum_to_xyz(spectrum):
    s = np.ascontiguousarray(spectrum, dtype=np.float32)
    out = (C.c_float * 3)()
    L.check(L.lib().rtmi_spectrum_to_xyz(s.ctypes.data_as(C.POINTER(C.c_float)), out))
    return np.array(list(out), dtype=np.float32)


def cie_tables():
    x, y, z = ((C.c_float * L.NBINS)() for _ in range(3))
    L.check(L.lib().rtmi_spectrum_cie_tables(x, y, z))
    return np.array([list(x), list(y), list(z)], dtype=np.float32)


def camera_new(focus, focal_distance, field_of_view, rotation, width, height):
    """Camera::new (src/camera.rs:19-51)."""
    cam = L.Camera()
    L.check(L.lib().rtmi_camera_new((C.c_double * 3)(*focus), float(focal_distance), float(field_of_view),
                                    (C.c_double * 3)(*rotation), int(width), int(height), C.byref(cam)))
    return cam


def camera_get_ray(cam, i, j):
    r = L.Ray()
    L.check(L.lib().rtmi_camera_get_ray(C.byref(cam), float(i), float(j), C.byref(r)))
    return np.array(list(r.o)), np.array(list(r.d))


def render_params(integrator, max_depth, spp, seed=0, flags=0, sample_range=None, tile_rank=0, tile_world=0, tile_size=0):
    p = L.RenderParams(int(integrator), int(max_depth), int(spp), int(flags), int(seed), 0, 0, int(tile_rank),
                       int(tile_world), int(tile_size), 0)
    if sample_range is not None:
        p.sample_begin, p.sample_end = int(sample_range[0]), int(sample_range[1])
    return p


def stats_dict(st):
    return {k: getattr(st, k) for k, _ in L.Stats._fields_}


class Context:
    """One GPU. rtmi_ctx_create fails with RTMI_NO_DEVICE when no GPU is visible."""

    def __init__(self, device=-1, max_batch_paths=0):
        h = C.c_void_p()
        L.check(L.lib().rtmi_ctx_create(int(device), int(max_batch_paths), C.byref(h)))
        self.handle = h
        self._scenes = weakref.WeakSet()

    def close(self):
        """Destroys the scenes that still live on this context, then the context."""
        if getattr(self, "handle", None):
            for scene in list(self._scenes):
                scene.close()
            L.lib().rtmi_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: module globals may already be gone; the process frees the GPU
            pass

    def tonemap(self, xyz_sum_ptr, spp, width, height, rgb8_ptr, stream=0):
        L.check(L.lib().rtmi_tonemap(self.handle, C.c_void_p(xyz_sum_ptr), int(spp), int(width), int(height),
                                     C.c_void_p(rgb8_ptr), C.c_void_p(stream)))


class Scene:
    """A flattened scene resident on the context's GPU (rtmi_scene_create)."""

    def __init__(self, ctx, objects, spheres, triangles, materials, hit_rule=L.HIT_FIRST_INDEX, flags=0):
        self.ctx = ctx
        objects = np.ascontiguousarray(objects, dtype=L.OBJECT_DT)
        spheres = np.ascontiguousarray(spheres, dtype=L.SPHERE_DT)
        triangles = np.ascontiguousarray(triangles, dtype=L.TRIANGLE_DT)
        materials = np.ascontiguousarray(materials, dtype=L.MATERIAL_DT)
        d = L.SceneDesc(objects.ctypes.data, len(objects), spheres.ctypes.data, len(spheres), triangles.ctypes.data,
                        len(triangles), materials.ctypes.data, len(materials), int(hit_rule), int(flags))
        h = C.c_void_p()
        L.check(L.lib().rtmi_scene_create(ctx.handle, C.byref(d), C.byref(h)))
        self.handle = h
        ctx._scenes.add(self)

    def close(self):
        if getattr(self, "handle", None):
            L.lib().rtmi_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: module globals may already be gone
            pass

    def info(self):
        i = L.SceneInfo()
        L.check(L.lib().rtmi_scene_get_info(self.handle, C.byref(i)))
        return {k: getattr(i, k) for k, _ in L.SceneInfo._fields_}

    def intersect_batch(self, rays, want_stats=False):
        rays = np.ascontiguousarray(rays, dtype=L.RAY_DT)
        hits = np.zeros(len(rays), dtype=L.HIT_DT)
        st = L.Stats()
        L.check(L.lib().rtmi_intersect_batch(self.handle, rays.ctypes.data, len(rays), hits.ctypes.data,
                                             C.byref(st) if want_stats else None))
        return (hits, stats_dict(st)) if want_stats else hits

    def intersect_batch_device(self, rays_ptr, n, hits_ptr, stream=0):
        L.check(L.lib().rtmi_intersect_batch_device(self.handle, C.c_void_p(rays_ptr), int(n), C.c_void_p(hits_ptr),
                                                    C.c_void_p(stream)))

    def trace_batch(self, rays, integrator, max_depth, seed=0, pixel_base=0, sample_index=0):
        rays = np.ascontiguousarray(rays, dtype=L.RAY_DT)
        out = np.zeros((len(rays), L.NBINS), dtype=np.float32)
        st = L.Stats()
        L.check(L.lib().rtmi_trace_batch(self.handle, rays.ctypes.data, len(rays), int(integrator), int(max_depth),
                                         int(seed), int(pixel_base), int(sample_index), out.ctypes.data, C.byref(st)))
        return out, stats_dict(st)

    def render(self, cam, params, want_xyz=True):
        w, h = cam.width, cam.height
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        xyz = np.zeros((h, w, 3), dtype=np.float32) if want_xyz else None
        st = L.Stats()
        L.check(L.lib().rtmi_render(self.handle, C.byref(cam), C.byref(params), rgb8.ctypes.data,
                                    xyz.ctypes.data if want_xyz else None, C.byref(st)))
        return rgb8, xyz, stats_dict(st)

    def render_accumulate(self, cam, params, xyz_sum_ptr, stream=0):
        """Adds this call's share of the samples into a device buffer of W*H*3 floats (asynchronous)."""
        L.check(L.lib().rtmi_render_accumulate(self.handle, C.byref(cam), C.byref(params), C.c_void_p(xyz_sum_ptr),
                                               C.c_void_p(stream), None))

    def render_accumulate_noalloc(self, cam, params, xyz_sum_ptr, stream=0):
        """render_accumulate in the library's static scratch array: no device allocation (asynchronous)."""
        L.check(L.lib().rtmi_render_accumulate_noalloc(self.handle, C.byref(cam), C.byref(params), C.c_void_p(xyz_sum_ptr),
                                                       C.c_void_p(stream)))

    def collect_stats(self, stream=0):
        st = L.Stats()
        L.check(L.lib().rtmi_render_collect_stats(self.handle, C.c_void_p(stream), C.byref(st)))
        return stats_dict(st)


class MultiScene:
    """A scene replicated over several GPUs of this process (rtmi_multi_create): one render is partitioned over them
    and merged with one collective."""

    def __init__(self, device_ids, objects, spheres, triangles, materials, hit_rule=L.HIT_FIRST_INDEX, max_batch_paths=0):
        objects = np.ascontiguousarray(objects, dtype=L.OBJECT_DT)
        spheres = np.ascontiguousarray(spheres, dtype=L.SPHERE_DT)
        triangles = np.ascontiguousarray(triangles, dtype=L.TRIANGLE_DT)
        materials = np.ascontiguousarray(materials, dtype=L.MATERIAL_DT)
        d = L.SceneDesc(objects.ctypes.data, len(objects), spheres.ctypes.data, len(spheres), triangles.ctypes.data,
                        len(triangles), materials.ctypes.data, len(materials), int(hit_rule), 0)
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        L.check(L.lib().rtmi_multi_create(ids, len(device_ids), int(max_batch_paths), C.byref(d), C.byref(h)))
        self.handle, self.n = h, len(device_ids)

    def close(self):
        if getattr(self, "handle", None):
            L.lib().rtmi_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = L.MultiInfo()
        L.check(L.lib().rtmi_multi_info(self.handle, C.byref(i)))
        return {"n_ranks": i.n_ranks, "uses_rccl": bool(i.uses_rccl), "rccl_library": i.rccl_library.decode()}

    def render(self, cam, params, partition=L.PARTITION_SAMPLES, want_xyz=True):
        w, h = cam.width, cam.height
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        xyz = np.zeros((h, w, 3), dtype=np.float32) if want_xyz else None
        st = L.Stats()
        per = (L.Stats * self.n)()
        L.check(L.lib().rtmi_multi_render(self.handle, C.byref(cam), C.byref(params), int(partition), rgb8.ctypes.data,
                                          xyz.ctypes.data if want_xyz else None, C.byref(st), per))
        return rgb8, xyz, stats_dict(st), [stats_dict(p) for p in per]

    def render_pass(self, cam, params, partition, pass_index, n_passes, want_xyz=True):
        """One pass of a progressive render; returns the image of all the samples taken so far."""
        w, h = cam.width, cam.height
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        xyz = np.zeros((h, w, 3), dtype=np.float32) if want_xyz else None
        st = L.Stats()
        L.check(L.lib().rtmi_multi_render_pass(self.handle, C.byref(cam), C.byref(params), int(partition), int(pass_index), int(n_passes),
                                               rgb8.ctypes.data, xyz.ctypes.data if want_xyz else None, C.byref(st)))
        return rgb8, xyz, stats_dict(st)
