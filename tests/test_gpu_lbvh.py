"""RTMI_SCENE_GPU_BVH: the meshes' trees built on the GPU (Morton order + binary radix tree) instead of the host's SAH
build. A tree only prunes work, so every hit record, path and image must be what the oracle's linear scan gives -- and
what the host-built tree gives -- under both hit rules, from the 12-triangle cube to the 1 048 576-triangle mesh."""
import numpy as np
import pytest

import _oracle as O
import _scenes as S
from raytracing_amd import scenes, _lib as L
from test_gpu_parity import RECIPES, check_hits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from raytracing_amd import api
    c = api.Context(0, max_batch_paths=1 << 22)
    yield c
    c.close()


@pytest.mark.parametrize("name", ["scene1", "scene2", "mesh"])
@pytest.mark.parametrize("rule", [O.HIT_FIRST_INDEX, O.HIT_NEAREST])
def test_hits_with_the_device_built_tree(ctx, name, rule):
    recipe, kw = RECIPES[name]
    osc, ocam, _ = S.oracle_scene(recipe, hit_rule=rule, **kw)
    gsc, _, _ = S.library_scene(ctx, recipe, hit_rule=rule, flags=L.SCENE_GPU_BVH, **kw)
    info = gsc.info()
    assert info["bvh_device_ms"] > 0.0 and info["n_nodes"] >= 1
    rays = np.concatenate([S.primary_rays(ocam), S.random_rays(1 << 15, seed=5, center=(0.0, 0.0, 0.5), radius=2.8 if name == "scene1" else 6.0, spread=1.5)])
    gpu, st = gsc.intersect_batch(rays, want_stats=True)
    assert check_hits(gpu, osc.intersect_batch(rays), "%s, device-built tree, rule %d" % (name, rule)) == 1.0
    assert st["nodes_visited"] > 0 and st["prims_tested"] > 0


def test_multi_mesh_render_with_device_built_trees(ctx):
    from raytracing_amd import api
    for rule in (O.HIT_FIRST_INDEX, O.HIT_NEAREST):
        osc, ocam, _ = S.oracle_scene(S.multi_mesh_scene, hit_rule=rule)
        host_tree, cam, _ = S.library_scene(ctx, S.multi_mesh_scene, hit_rule=rule)
        dev_tree, _, _ = S.library_scene(ctx, S.multi_mesh_scene, hit_rule=rule, flags=L.SCENE_GPU_BVH)
        p = api.render_params(O.INT_RANDOM_WALK, 10, 6, seed=6)
        i1, x1, s1 = host_tree.render(cam, p)
        i2, x2, s2 = dev_tree.render(cam, p)
        assert (x1.view(np.uint32) == x2.view(np.uint32)).all() and (i1 == i2).all() and s1["segments"] == s2["segments"]
        ref8, _, ref_lin, ref_st = osc.render(ocam, O.INT_RANDOM_WALK, 10, 6, seed=6)
        assert (x2 == ref_lin).all(axis=2).mean() >= 0.999 and abs(s2["segments"] - ref_st.segments) <= 1e-4 * ref_st.segments + 2


def test_degenerate_inputs(ctx):
    """Two triangles, coincident triangles (equal Morton codes) and a flat mesh."""
    from raytracing_amd import api
    from raytracing_amd.scene_builder import SceneBuilder
    tri = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0])
    meshes = [np.stack([tri, tri + 0.5]), np.stack([tri] * 9), np.stack([tri + np.tile([0.1 * k, 0.05 * k, 0.0], 3) for k in range(40)])]
    rays = S.random_rays(4000, seed=9, center=(0.5, 0.5, 0.2), radius=3.0, spread=1.0)
    for mesh in meshes:
        for rule in (O.HIT_FIRST_INDEX, O.HIT_NEAREST):
            bo, bg = SceneBuilder(O.from_rgb), SceneBuilder()
            bo.add_model(mesh); bg.add_model(mesh)
            ref = O.Scene(*bo.arrays(), hit_rule=rule).intersect_batch(rays)
            gpu = api.Scene(ctx, *bg.arrays(), hit_rule=rule, flags=L.SCENE_GPU_BVH).intersect_batch(rays)
            assert ((gpu["obj"] == ref["obj"]) & (gpu["prim"] == ref["prim"])).all()
            assert (ref["obj"] != O.MISS).any()


def test_million_triangle_mesh_both_builders_agree(ctx):
    from raytracing_amd import api
    b, cam_args = scenes.synthetic_mesh_scene(from_rgb=None, width=3840, height=2160)
    arr = b.arrays()
    rays = np.concatenate([S.primary_rays(O.camera(*cam_args), step=31), S.random_rays(1 << 16, seed=4, center=(0.0, 0.0, 1.0), radius=3.0, spread=1.1)])
    for rule in (O.HIT_NEAREST, O.HIT_FIRST_INDEX):
        host_tree = api.Scene(ctx, *arr, hit_rule=rule)
        dev_tree = api.Scene(ctx, *arr, hit_rule=rule, flags=L.SCENE_GPU_BVH)
        a, b_ = host_tree.intersect_batch(rays), dev_tree.intersect_batch(rays)
        assert a.tobytes() == b_.tobytes() and (a["obj"] == 0).sum() > 10000
        hi, di = host_tree.info(), dev_tree.info()
        print("rule %d: host SAH build %.0f ms, device build %.0f ms (%.0f ms on the device side), nodes %d vs %d, depth %d vs %d"
              % (rule, hi["bvh_build_ms"], di["bvh_build_ms"], di["bvh_device_ms"], hi["n_nodes"], di["n_nodes"], hi["max_depth"], di["max_depth"]))
        assert di["bvh_build_ms"] < hi["bvh_build_ms"]
        host_tree.close(); dev_tree.close()
