"""CPU-side checks of the product library: it loads, exports exactly what include/rtmi.h declares,
its plain structs have the documented sizes, and its host-side helpers (spectra, camera) agree bit
for bit with the oracle. No kernel is launched here."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def L():
    from raytracing_amd import _lib
    _lib.lib()
    return _lib


def header_functions():
    text = (ROOT / "include" / "rtmi.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtmi_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(L):
    declared = header_functions()
    assert len(declared) >= 18
    lib = L.lib()
    for name in declared:
        assert hasattr(lib, name), "librtmi.so does not export %s" % name
        assert name in L.SYMBOLS, "%s has no ctypes prototype" % name
    assert sorted(L.SYMBOLS) == declared


def test_struct_layouts(L):
    assert L.RAY_DT.itemsize == 48 and L.HIT_DT.itemsize == 64
    assert L.OBJECT_DT.itemsize == 16 and L.SPHERE_DT.itemsize == 104 and L.TRIANGLE_DT.itemsize == 72
    assert L.MATERIAL_DT.itemsize == 8 + 4 * 60 * 4 + 8
    assert C.sizeof(L.Camera) == 15 * 8 + 8
    assert C.sizeof(L.RenderParams) == 48
    assert C.sizeof(L.Stats) == 80
    assert C.sizeof(L.SceneDesc) == 72


def test_version_and_error_string(L):
    assert L.lib().rtmi_version().decode().startswith("rtmi ")
    out = (C.c_float * 60)()
    assert L.lib().rtmi_spectrum_from_rgb(C.c_float(0.1), C.c_float(0.2), C.c_float(0.3), 7, out) == L.BAD_ARG
    assert b"spectrum type" in L.lib().rtmi_last_error()


def test_no_cpu_fallback(L):
    """Without a GPU the library refuses to create a context instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    assert L.lib().rtmi_ctx_create(0, 0, C.byref(h)) == L.NO_DEVICE
    assert not h.value


COLOURS = [(0.8, 0.2, 0.2), (1.0, 1.0, 1.0), (0.376, 0.239, 0.141), (0.2, 0.1, 0.9), (0.5, 0.5, 0.5), (0.1, 0.9, 0.3),
           (0.0, 0.0, 0.0), (0.3, 0.3, 0.9), (0.9, 0.3, 0.3), (0.2, 0.9, 0.9), (0.6, 0.5, 0.1), (1.5, 0.2, 0.7)]


@pytest.mark.parametrize("kind", [0, 1])
def test_spectra_match_oracle_bitwise(oracle, kind):
    from raytracing_amd import api
    for rgb in COLOURS:
        a, b = api.spectrum_from_rgb(rgb, kind), oracle.from_rgb(rgb, kind)
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), rgb
        assert (api.spectrum_to_xyz(a).view(np.uint32) == oracle.to_xyz(b).view(np.uint32)).all(), rgb


def test_cie_tables_match_oracle_bitwise(oracle):
    from raytracing_amd import api
    x, y, z = ((C.c_float * 60)() for _ in range(3))
    oracle.lib().ref_spectrum_tables(x, y, z)
    ref = np.array([list(x), list(y), list(z)], dtype=np.float32)
    assert (api.cie_tables().view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.parametrize("args", [
    ((0, 0, 0), 1.0, 90.0, (0, 0, 0), 500, 1000),
    ((0, 0, 0), 1.0, 90.0, (np.pi / 2, 0, 0), 500, 1000),
    ((-2.5, 0.0, 2.0), 0.1, 120.0, (0.0, 0.5, 0.0), 400, 225),
    ((1.0, -3.0, 0.5), 0.35, 60.0, (0.2, -0.4, 1.1), 1920, 1080),
])
def test_camera_matches_oracle_bitwise(oracle, args):
    from raytracing_amd import api
    a, b = api.camera_new(*args), oracle.camera(*args)
    for f in ("focus", "p1", "p2", "p3", "p4"):
        assert list(getattr(a, f)) == list(getattr(b, f)), f
    r = oracle.Ray()
    for i, j in ((0.0, 0.0), (3.25, 7.5), (args[4] - 0.5, args[5] - 0.5)):
        o, d = api.camera_get_ray(a, i, j)
        oracle.lib().ref_camera_get_ray(C.byref(b), i, j, C.byref(r))
        assert list(o) == list(r.o) and list(d) == list(r.d)


def test_camera_known_answers():
    """The reference's own camera tests (src/camera.rs:94-163), through the library."""
    from raytracing_amd import api
    c = api.camera_new((0, 0, 0), 1.0, 90.0, (0, 0, 0), 500, 1000)
    assert np.allclose(list(c.p1), (1, -1, 2), atol=1e-10) and np.allclose(list(c.p4), (1, 1, -2), atol=1e-10)
    c = api.camera_new((0, 0, 0), 1.0, 90.0, (np.pi / 2, 0, 0), 500, 1000)
    assert np.allclose(list(c.p1), (1, -2, -1), atol=1e-10) and np.allclose(list(c.p4), (1, 2, 1), atol=1e-10)
    c = api.camera_new((0, 0, 0), 1.0, 90.0, (0, 0, 0), 100, 100)
    _, d = api.camera_get_ray(c, 0.0, 0.0)
    assert np.allclose(d, np.array([1, -1, 1]) / np.sqrt(3), atol=1e-10)
    _, d = api.camera_get_ray(c, 50.0, 50.0)
    assert np.allclose(d, (1, 0, 0), atol=1e-10)
