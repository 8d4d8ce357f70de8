"""The C++ loaders on the reference's own asset files (scene.toml, scene2.toml, scene3.toml, cubo.obj/.mtl,
mono.obj/.mtl), read where they lie in the build container. The GPU box has no /root/reference: there these tests are
skipped and the committed fixtures (raytracing_amd/data/meshes.npz, the recipes in raytracing_amd/scenes.py) stand in, which is
why this file checks that those stand-ins say exactly what the files say. Nothing is copied: the files are only parsed."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest

import _host as H
from raytracing_amd import api, scenes

REF = Path("/root/reference")
pytestmark = pytest.mark.skipif(not (REF / "scene.toml").exists(), reason="the reference checkout is not present on this machine")


def same_arrays(a, b):
    return all(x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes() for x, y in zip(a, b))


@pytest.mark.parametrize("toml,recipe", [("scene.toml", scenes.scene1), ("scene2.toml", scenes.scene2), ("scene3.toml", scenes.scene3)])
def test_reference_scene_files_equal_the_recipes(monkeypatch, toml, recipe):
    # the reference opens OBJ and MTL paths relative to the working directory (src/shapes/model_obj.rs:16,21)
    monkeypatch.chdir(REF)
    arrays, cam = H.LoadedScene(REF / toml).arrays()
    b, cam_args = recipe()
    assert same_arrays(arrays, b.arrays()), toml
    assert bytes(cam) == bytes(api.camera_new(*cam_args)), toml


def test_mesh_fixture_regenerates_identically(tmp_path):
    spec = importlib.util.spec_from_file_location("make_mesh_fixtures", Path(__file__).resolve().parent / "golden" / "make_mesh_fixtures.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    committed = np.load(Path(__file__).resolve().parent.parent / "raytracing_amd" / "data" / "meshes.npz")
    for name, fname in (("cubo", "cubo.obj"), ("mono", "mono.obj")):
        v, i, o = mod.read_obj(REF / fname)
        assert (committed[name + "_vertices"] == v).all() and committed[name + "_vertices"].dtype == v.dtype
        assert (committed[name + "_face_index"] == i).all() and (committed[name + "_face_offsets"] == o).all()
    # Suzanne: 507 vertices, 468 quads + 32 triangles = 968 triangles; the cube: 8 vertices, 6 quads
    assert len(committed["mono_vertices"]) == 507 and len(committed["mono_face_offsets"]) - 1 == 500
    assert len(committed["cubo_vertices"]) == 8 and len(committed["cubo_face_offsets"]) - 1 == 6
