"""The workloads the benchmark and the parity tests run, as SceneBuilder recipes.

The three stock scenes restate the numbers of the reference's scene files (scene.toml,
scene2.toml, scene3.toml) and mesh materials (cubo.mtl, mono.mtl); the synthetic mesh scene and
the deep-bounce scene are the larger configurations of the benchmark plan. Every function returns
(builder, camera_args) where camera_args = (focus, focal_distance, fov_deg, rotation, width, height).
"""
import numpy as np

from .scene_builder import SceneBuilder, load_golden_mesh, procedural_mesh


def _cubo_material(b):
    # cubo.mtl:4-12 -- Ns 360, Ka 1 1 1, Kd .162632 .037185 .010680, Ks .5 .5 .5, Ke 0 0 0 (Ke present), Ni 1.45
    return b.mtl_material((1.0, 1.0, 1.0), (0.162632, 0.037185, 0.010680), (0.5, 0.5, 0.5), (0.0, 0.0, 0.0), ns=360.0, ni=1.45)


def _mono_material(b):
    # mono.mtl:4-10 -- Ns 500, Ka = Kd = Ks = 0.8, no Ke, no Ni
    return b.mtl_material((0.8, 0.8, 0.8), (0.8, 0.8, 0.8), (0.8, 0.8, 0.8), None, ns=500.0)


def scene1(from_rgb=None, width=400, height=225, fan="last"):
    """scene.toml: the cube room with four spheres and two triangles; camera inside the cube."""
    b = SceneBuilder(from_rgb)
    b.add_model(load_golden_mesh("cubo", fan), _cubo_material(b))
    b.add_sphere((0.0, -2.0, 1.0), 1.0, material=b.material("Lambertian", (0.8, 0.2, 0.2)))
    b.add_sphere((0.0, 0.0, 1.0), 1.0, material=b.material("Specular", (0.8, 0.8, 0.8)))
    b.add_sphere((0.0, 1.5, 1.5), 1.0, scale=(1.0, 0.5, 1.5), material=b.material("Specular", (0.8, 0.2, 0.2)))
    b.add_triangle([[-200.0, -100.0, 0.0], [100.0, 200.0, 0.0], [100.0, -100.0, 0.0]],
                   material=b.material("Lambertian", (0.6, 0.5, 0.1)))
    b.add_triangle([[-1.0, 1.0, 0.01], [-1.0, 2.0, 0.01], [0.0, 2.0, 0.01]], material=b.material("Specular", (0.2, 0.1, 0.9)))
    b.add_sphere((0.0, 0.0, 3.0), 0.8, material=b.material("Emitter", (0.6, 0.6, 0.6)))
    return b, ((-2.5, 0.0, 2.0), 0.1, 120.0, (0.0, 0.5, 0.0), width, height)


def scene2(from_rgb=None, width=850, height=480, fan="last", specular_mesh=False):
    """scene2.toml: Suzanne (968 triangles) under a white emitter sphere on a r = 999.5 ground sphere.
    specular_mesh=True swaps the mesh material for a 0.8 mirror (an extension: the reference's
    ModelObj loader cannot override the MTL material)."""
    b = SceneBuilder(from_rgb)
    mat = b.material("Specular", (0.8, 0.8, 0.8)) if specular_mesh else _mono_material(b)
    b.add_model(load_golden_mesh("mono", fan), mat)
    b.add_sphere((0.0, -4.0, 4.0), 1.5, material=b.material("Emitter", (1.0, 1.0, 1.0)))
    b.add_sphere((0.0, 0.0, -1000.0), 999.5, material=b.material("Lambertian", (0.376, 0.239, 0.141)))
    return b, ((-2.5, 0.0, 2.0), 0.1, 120.0, (0.0, 0.5, 0.0), width, height)


def scene3(from_rgb=None, width=1280, height=720):
    """scene3.toml: six spheres (emitter, Lambertian, three mirrors, r = 1000 ground)."""
    b = SceneBuilder(from_rgb)
    b.add_sphere((0.0, 0.0, 3.0), 1.0, material=b.material("Emitter", (1.0, 1.0, 1.0)))
    b.add_sphere((0.0, 0.0, 1.0), 1.0, material=b.material("Lambertian", (0.8, 0.3, 0.2)))
    for c in ((0.0, 1.3, 0.5), (0.0, -1.3, 0.5), (-1.3, 0.0, 0.5)):
        b.add_sphere(c, 0.5, material=b.material("Specular", (0.8, 0.8, 0.8)))
    b.add_sphere((0.0, 0.0, -1000.0), 1000.0, material=b.material("Lambertian", (0.208, 0.141, 0.094)))
    return b, ((-4.5, 0.0, 2.0), 0.1, 120.0, (0.0, 0.5, 0.0), width, height)


def scene3_extended(from_rgb=None, width=1280, height=720):
    """scene3's layout with the materials of its five small spheres cycled through mirror, Lambertian, glossy
    (Ns 50 and 1000) and dielectric (Ni 1.5). Glossy and dielectric are extensions: no reference counterpart."""
    b = SceneBuilder(from_rgb)
    b.add_sphere((0.0, 0.0, 3.0), 1.0, material=b.material("Emitter", (1.0, 1.0, 1.0)))
    b.add_sphere((0.0, 0.0, 1.0), 1.0, material=b.material("Dielectric", (0.95, 0.95, 0.95), ni=1.5))
    mats = [b.material("Glossy", (0.8, 0.8, 0.8), ns=50.0), b.material("Glossy", (0.8, 0.6, 0.3), ns=1000.0),
            b.material("Specular", (0.8, 0.8, 0.8))]
    for c, m in zip(((0.0, 1.3, 0.5), (0.0, -1.3, 0.5), (-1.3, 0.0, 0.5)), mats):
        b.add_sphere(c, 0.5, material=m)
    b.add_sphere((-1.6, 1.1, 0.35), 0.35, material=b.material("Lambertian", (0.2, 0.5, 0.3)))
    b.add_sphere((0.0, 0.0, -1000.0), 1000.0, material=b.material("Lambertian", (0.208, 0.141, 0.094)))
    return b, ((-4.5, 0.0, 2.0), 0.1, 120.0, (0.0, 0.5, 0.0), width, height)


def synthetic_mesh_scene(from_rgb=None, n_lat=512, n_lon=1024, width=3840, height=2160):
    """Procedural closed mesh (2*n_lat*n_lon triangles) in the scene3 surroundings."""
    b = SceneBuilder(from_rgb)
    b.add_model(procedural_mesh(n_lat, n_lon), b.material("Lambertian", (0.7, 0.7, 0.7)))
    b.add_sphere((0.0, 0.0, 3.5), 1.0, material=b.material("Emitter", (1.0, 1.0, 1.0)))
    b.add_sphere((0.0, 0.0, -1000.0), 1000.0, material=b.material("Lambertian", (0.208, 0.141, 0.094)))
    return b, ((-4.5, 0.0, 2.0), 0.1, 120.0, (0.0, 0.5, 0.0), width, height)


def primary_rays(camera_get_ray, cam_args, step=1):
    """Rays through the pixel centres (jitter 0) of a camera, as (n,) records of o[3], d[3]."""
    _, _, _, _, w, h = cam_args
    xs, ys = np.arange(0, w, step), np.arange(0, h, step)
    rays = np.zeros(len(xs) * len(ys), dtype=[("o", "<f8", 3), ("d", "<f8", 3)])
    k = 0
    for y in ys:
        for x in xs:
            o, d = camera_get_ray(float(x), float(y))
            rays["o"][k], rays["d"][k] = o, d
            k += 1
    return rays
