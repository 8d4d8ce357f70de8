"""Builds the flat scene arrays of the C ABI with the semantics of the reference's loaders.

Mirrors src/scene_config.rs (Sphere/Triangle/ModelObj/Material::from_toml), src/material.rs
(Default, From<&mtl::Material>) and the geometry half of src/shapes/model_obj.rs. The spectrum
conversion is injectable so that tests can build the same scene once through the library and
once through the CPU oracle.
"""
from pathlib import Path

import numpy as np

from . import _lib as L

# geometry of the two stock meshes (cube room, Suzanne) as polygon soups; written by tests/golden/make_mesh_fixtures.py
GOLDEN_MESHES = Path(__file__).resolve().parent / "data" / "meshes.npz"


def fan_triangulate(vertices, face_index, face_offsets, origin="last"):
    """Polygon faces -> (n, 9) triangle array, winding preserved, face order kept.

    origin="last": polygon (v0..vn) -> (vn, v0, v1), (vn, v1, v2), ...: the fan the OBJ crate the
    reference uses (wavefront_obj 10.0) is believed to emit; origin="first": (v0, v1, v2), (v0, v2, v3), ...
    No reference test pins this order (parity unpinned); it only changes which triangle index wins
    under the first-index hit rule.
    """
    tris = []
    for f in range(len(face_offsets) - 1):
        idx = face_index[face_offsets[f]:face_offsets[f + 1]]
        n = len(idx)
        if n < 3:
            continue
        if origin == "last":
            for i in range(n - 2):
                tris.append((idx[n - 1], idx[i], idx[i + 1]))
        else:
            for i in range(1, n - 1):
                tris.append((idx[0], idx[i], idx[i + 1]))
    tris = np.asarray(tris, dtype=np.int64)
    return np.ascontiguousarray(vertices[tris].reshape(-1, 9), dtype=np.float64)


def load_golden_mesh(name, origin="last", path=GOLDEN_MESHES):
    z = np.load(path)
    return fan_triangulate(z[name + "_vertices"], z[name + "_face_index"], z[name + "_face_offsets"], origin)


def _hash_u32(x):
    x = x.astype(np.uint64)
    x = (x ^ (x >> np.uint64(16))) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(15))) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)
    return (x ^ (x >> np.uint64(16))) & np.uint64(0xFFFFFFFF)


def _value_noise(p, seed):
    """Trilinear value noise on the integer lattice from an integer hash (deterministic)."""
    pf = np.floor(p)
    fr = p - pf
    pi = pf.astype(np.int64)
    w = fr * fr * (3.0 - 2.0 * fr)
    out = np.zeros(len(p))
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                key = ((pi[:, 0] + dx) * 73856093) ^ ((pi[:, 1] + dy) * 19349663) ^ ((pi[:, 2] + dz) * 83492791) ^ seed
                h = _hash_u32(np.asarray(key) & 0xFFFFFFFF).astype(np.float64) / 4294967296.0
                wx = w[:, 0] if dx else 1.0 - w[:, 0]
                wy = w[:, 1] if dy else 1.0 - w[:, 1]
                wz = w[:, 2] if dz else 1.0 - w[:, 2]
                out += h * wx * wy * wz
    return out


def procedural_mesh(n_lat=512, n_lon=1024, center=(0.0, 0.0, 1.0), radius=1.0, amplitude=0.15, seed=0x5EED):
    """Closed lat-long sphere of 2*n_lat*n_lon triangles (degenerate ones at the poles included),
    displaced radially by 1 + amplitude * fbm(p). Outward winding. 512 x 1024 -> 1 048 576 triangles."""
    lat = np.linspace(0.0, np.pi, n_lat + 1)
    lon = np.linspace(0.0, 2.0 * np.pi, n_lon + 1)[:-1]
    st, ct = np.sin(lat)[:, None], np.cos(lat)[:, None]
    d = np.stack([st * np.cos(lon)[None, :], st * np.sin(lon)[None, :], np.broadcast_to(ct, (n_lat + 1, n_lon))], axis=-1)
    d[0, :, :] = (0.0, 0.0, 1.0)
    d[-1, :, :] = (0.0, 0.0, -1.0)
    flat = d.reshape(-1, 3)
    fbm = np.zeros(len(flat))
    freq, amp = 2.0, 0.5
    for octave in range(4):
        fbm += amp * (_value_noise(flat * freq + 11.5, seed + octave) * 2.0 - 1.0)
        freq *= 2.0
        amp *= 0.5
    r = radius * (1.0 + amplitude * fbm)
    pts = (flat * r[:, None] + np.asarray(center)[None, :]).reshape(n_lat + 1, n_lon, 3)
    i0 = np.arange(n_lat)[:, None]
    j0 = np.arange(n_lon)[None, :]
    j1 = (j0 + 1) % n_lon
    a, b = pts[i0, j0], pts[i0 + 1, j0]
    c, e = pts[i0 + 1, j1], pts[i0, j1]
    t1 = np.concatenate([a, b, c], axis=-1)
    t2 = np.concatenate([a, c, e], axis=-1)
    return np.ascontiguousarray(np.stack([t1, t2], axis=2).reshape(-1, 9))


class SceneBuilder:
    def __init__(self, from_rgb=None):
        if from_rgb is None:
            from .api import spectrum_from_rgb as from_rgb
        self.from_rgb = from_rgb
        self.objects, self.spheres, self.materials = [], [], []
        self._tri_blocks, self._n_tris = [], 0

    # ---- materials ------------------------------------------------------------------------
    def _material(self, type_, ambient=None, emitted=None, diffused=None, specular=None, ns=0.0, ni=1.0):
        m = np.zeros((), dtype=L.MATERIAL_DT)
        m["type"] = type_
        m["specular_coefficient"], m["optical_density"] = ns, ni
        for bit, name, spec in ((L.HAS_AMBIENT, "ambient", ambient), (L.HAS_EMITTED, "emitted", emitted),
                                (L.HAS_DIFFUSED, "diffused", diffused), (L.HAS_SPECULAR, "specular", specular)):
            if spec is not None:
                m["present"] |= bit
                m[name] = spec
        self.materials.append(m)
        return len(self.materials) - 1

    def default_material(self):
        """Material::default(): an Emitter with no colour, i.e. a black absorber (src/material.rs:25-37)."""
        return self._material(L.MAT_EMITTER)

    def material(self, type_, albedo, ns=0.0, ni=1.0):
        """Material::from_toml (src/scene_config.rs:197-250); Emitter albedo uses the illuminant basis.
        "Glossy" (exponent ns) and "Dielectric" (index ni) are extensions with no counterpart in the reference."""
        if type_ == "Glossy":
            return self._material(L.MAT_GLOSSY, specular=self.from_rgb(albedo, L.SPEC_REFLECTANCE), ns=ns)
        if type_ == "Dielectric":
            return self._material(L.MAT_DIELECTRIC, specular=self.from_rgb(albedo, L.SPEC_REFLECTANCE), ni=ni)
        if type_ == "Lambertian":
            return self._material(L.MAT_LAMBERTIAN, ambient=self.from_rgb(albedo, L.SPEC_REFLECTANCE))
        if type_ == "Specular":
            return self._material(L.MAT_SPECULAR, specular=self.from_rgb(albedo, L.SPEC_REFLECTANCE))
        if type_ == "Emitter":
            return self._material(L.MAT_EMITTER, emitted=self.from_rgb(albedo, L.SPEC_ILLUMINANT))
        raise ValueError("unknown material type %r" % (type_,))

    def mtl_material(self, ka, kd, ks, ke=None, ns=0.0, ni=None):
        """From<&mtl::Material> (src/material.rs:39-53): always Lambertian, ambient = Ka; Ns and Ni are carried along
        (an absent Ni flattens to 1)."""
        refl = lambda c: self.from_rgb(c, L.SPEC_REFLECTANCE)
        return self._material(L.MAT_LAMBERTIAN, ambient=refl(ka), emitted=refl(ke) if ke is not None else None,
                              diffused=refl(kd), specular=refl(ks), ns=ns, ni=1.0 if ni is None else ni)

    # ---- shapes, in scene order -----------------------------------------------------------
    def add_sphere(self, center, radius, scale=(1.0, 1.0, 1.0), material=None):
        """Sphere::from_toml (src/scene_config.rs:128-166): local->world = T(center) * S(scale)."""
        s = np.zeros((), dtype=L.SPHERE_DT)
        s["l2w"] = [scale[0], 0, 0, center[0], 0, scale[1], 0, center[1], 0, 0, scale[2], center[2]]
        s["radius"] = radius
        self.spheres.append(s)
        self._object(L.OBJ_SPHERE, len(self.spheres) - 1, 1, material)

    def add_triangle(self, vertices, material=None):
        """Triangle::from_toml (src/scene_config.rs:170-192), identity transform."""
        t = np.zeros(1, dtype=L.TRIANGLE_DT)
        t["v"][0] = np.asarray(vertices, dtype=np.float64).reshape(9)
        self._tri_blocks.append(t)
        self._n_tris += 1
        self._object(L.OBJ_TRIANGLE, self._n_tris - 1, 1, material)

    def add_model(self, triangles9, material=None):
        """ModelObj (src/shapes/model_obj.rs:34-60): triangles in file order, one material."""
        tri = np.asarray(triangles9, dtype=np.float64).reshape(-1, 9)
        first = self._n_tris
        block = np.zeros(len(tri), dtype=L.TRIANGLE_DT)
        block["v"] = tri
        self._tri_blocks.append(block)
        self._n_tris += len(tri)
        self._object(L.OBJ_MODEL, first, len(tri), material)

    def _object(self, kind, first, count, material):
        if material is None:
            material = self.default_material()
        o = np.zeros((), dtype=L.OBJECT_DT)
        o["kind"], o["first"], o["count"], o["material"] = kind, first, count, material
        self.objects.append(o)

    def arrays(self):
        def stack(items, dt):
            return np.array(items, dtype=dt) if items else np.zeros(0, dtype=dt)
        tris = np.concatenate(self._tri_blocks) if self._tri_blocks else np.zeros(0, dtype=L.TRIANGLE_DT)
        return (stack(self.objects, L.OBJECT_DT), stack(self.spheres, L.SPHERE_DT), tris,
                stack(self.materials, L.MATERIAL_DT))
