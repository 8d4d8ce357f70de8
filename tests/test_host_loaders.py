"""The C++ host layer (raytracing_amd/host): TOML / OBJ / MTL loading with the reference's schema and
error behaviour (src/scene_config.rs, src/shapes/model_obj.rs, src/material.rs), checked against the
Python SceneBuilder recipes and against tomli. CPU only; the render path is in test_host_gpu.py."""
import numpy as np
import pytest
import tomli

import _host as H
from raytracing_amd import _lib as L, api, scenes


def same_arrays(a, b):
    return all(x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes() for x, y in zip(a, b))


def test_scene1_file_equals_recipe(tmp_path):
    loaded = H.LoadedScene(H.write_scene1(tmp_path, 400, 225))
    arrays, cam = loaded.arrays()
    b, cam_args = scenes.scene1()
    assert same_arrays(arrays, b.arrays())
    ref = api.camera_new(*cam_args)
    assert bytes(cam) == bytes(ref)


def test_scene2_file_equals_recipe_and_fan_option(tmp_path):
    path = H.write_scene2(tmp_path)
    arrays, cam = H.LoadedScene(path).arrays()
    b, cam_args = scenes.scene2()
    assert same_arrays(arrays, b.arrays())
    assert (cam.width, cam.height) == (850, 480)
    first, _ = H.LoadedScene(path, fan=1).arrays()
    assert same_arrays(first, scenes.scene2(fan="first")[0].arrays())
    assert first[2].tobytes() != arrays[2].tobytes()


def test_mesh_material_override_extension(tmp_path):
    arrays, _ = H.LoadedScene(H.write_scene2(tmp_path, specular=True)).arrays()
    assert same_arrays(arrays, scenes.scene2(specular_mesh=True)[0].arrays())


def test_only_first_mtl_material_and_ke_optional(tmp_path):
    arrays, _ = H.LoadedScene(H.write_scene1(tmp_path)).arrays()
    objects, _, _, mats = arrays
    cube = mats[objects["material"][0]]
    assert cube["type"] == L.MAT_LAMBERTIAN                      # always Lambertian, src/material.rs:42
    assert cube["present"] == L.HAS_AMBIENT | L.HAS_EMITTED | L.HAS_DIFFUSED | L.HAS_SPECULAR   # Ke 0 0 0 is still Some
    assert np.allclose(cube["ambient"], 1.0615, atol=2e-3)       # Ka = 1 1 1 through the Smits white basis
    arrays2, _ = H.LoadedScene(H.write_scene2(tmp_path)).arrays()
    monkey = arrays2[3][arrays2[0]["material"][0]]
    assert monkey["present"] == L.HAS_AMBIENT | L.HAS_DIFFUSED | L.HAS_SPECULAR


def test_toml_reader_agrees_with_tomli(tmp_path):
    text = H.SCENE1_TOML.format(w=40, h=30)
    doc = tomli.loads(text)
    (tmp_path / "cube.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    (tmp_path / "s.toml").write_text(text)
    arrays, cam = H.LoadedScene(tmp_path / "s.toml").arrays()
    objects, spheres, tris, mats = arrays
    assert (cam.width, cam.height) == (doc["Camera"]["width"], doc["Camera"]["height"])
    kinds = {"Sphere": L.OBJ_SPHERE, "Triangle": L.OBJ_TRIANGLE, "ModelObj": L.OBJ_MODEL}
    assert [kinds[e["type"]] for e in doc["Scene"]] == list(objects["kind"])
    k = 0
    for e in doc["Scene"]:
        if e["type"] == "Sphere":
            s = e.get("scale", [1.0, 1.0, 1.0])
            assert list(spheres["l2w"][k]) == [s[0], 0, 0, e["center"][0], 0, s[1], 0, e["center"][1], 0, 0, s[2], e["center"][2]]
            assert spheres["radius"][k] == e["radius"]
            k += 1
    tri_entries = [e for e in doc["Scene"] if e["type"] == "Triangle"]
    assert list(tris["v"][1]) == [c for p in tri_entries[0]["vertices"] for c in p]   # tris[0] is the one-triangle mesh
    assert objects["material"].tolist() == list(range(len(objects)))                  # no material on the mesh -> default


@pytest.mark.parametrize("mutate,message", [
    (lambda t: t.replace("[Camera]", "[Kamera]"), "No se ha especificado la cámara."),
    (lambda t: t.replace("width = 40", "width = 40.0"), "No se pudo cargar la cámara."),
    (lambda t: t.replace("field_of_view = 120.0", "field_of_view = 120"), "No se pudo cargar la cámara."),
    (lambda t: t.replace("radius = 0.8", "radius = 1"), "No se pudo cargar el modelo de esfera"),
    (lambda t: t.replace('type = "Emitter"', 'type = "Glass"'), "not yet implemented"),
    (lambda t: t.replace("albedo = [0.6, 0.6, 0.6]", "albedo = [0.6, 0.6]"), "No se pudo cargar el material"),
    (lambda t: t.replace("[[Scene]]", "[[Escena]]").replace("[Scene.material]", "[Escena.material]"), "No se ha especificado la escena."),
    (lambda t: t.replace('path = "cube.obj"', 'path = "missing.obj"'), "no encontrado"),
    (lambda t: t.replace("[0.0, 2.0, 0.01]]", "[0.0, 2.0, 0.01]"), "TOML parse error"),
])
def test_loader_errors_follow_the_reference(tmp_path, mutate, message):
    H.write_scene1(tmp_path, 40, 30)
    (tmp_path / "bad.toml").write_text(mutate(H.SCENE1_TOML.format(w=40, h=30)))
    with pytest.raises(H.LoadError) as e:
        H.LoadedScene(tmp_path / "bad.toml")
    assert message in str(e.value)


def test_unknown_shape_type_is_skipped(tmp_path):
    H.write_scene1(tmp_path, 40, 30)
    text = H.SCENE1_TOML.format(w=40, h=30) + '\n[[Scene]]\ntype = "Torus"\nradius = 2.0\n'
    (tmp_path / "t.toml").write_text(text)
    arrays, _ = H.LoadedScene(tmp_path / "t.toml").arrays()
    assert len(arrays[0]) == 7          # src/scene_config.rs:104-106: dbg!(s) and carry on


def test_bmp_writer(tmp_path):
    rgb = np.zeros((3, 5, 3), dtype=np.uint8)
    rgb[0, 0] = (255, 0, 0); rgb[2, 4] = (0, 0, 255); rgb[1, 2] = (10, 20, 30)
    assert H.host().rth_save_bmp(str(tmp_path / "a.bmp").encode(), rgb.ctypes.data, 5, 3) == 0
    from PIL import Image
    back = np.array(Image.open(tmp_path / "a.bmp").convert("RGB"))
    assert (back == rgb).all()
