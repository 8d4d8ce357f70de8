"""GPU parity: librtmi.so (HIP kernels, through the C ABI) against the CPU oracle.

Bars (from the project's north star): hit object/primitive ids equal on >= 99.99 % of rays and
hit t within 1e-5 relative; per-path spectra and images within the tolerances written below.
Geometry is binary64 on both sides; the differences that remain come from fused multiply-adds
and the GPU's sin/cos/pow, so the observed agreement is far tighter than the bars.
"""
import numpy as np
import pytest

import _oracle as O
import _scenes as S
from raytracing_amd import scenes

pytestmark = pytest.mark.gpu

T_REL = 1e-5          # hit distance, relative
ID_MATCH = 0.9999     # fraction of rays whose (object, primitive) ids must agree


@pytest.fixture(scope="module")
def ctx():
    from raytracing_amd import api
    c = api.Context(0, max_batch_paths=1 << 22)
    yield c
    c.close()


def check_hits(gpu, ref, what):
    assert len(gpu) == len(ref)
    same = (gpu["obj"] == ref["obj"]) & (gpu["prim"] == ref["prim"])
    frac = same.mean() if len(ref) else 1.0
    assert frac >= ID_MATCH, "%s: ids agree on %.6f of %d rays" % (what, frac, len(ref))
    hit = same & (ref["obj"] != O.MISS)
    if hit.any():
        rel = np.abs(gpu["t"][hit] - ref["t"][hit]) / np.maximum(np.abs(ref["t"][hit]), 1e-300)
        assert rel.max() <= T_REL, "%s: t differs by %.3e relative" % (what, rel.max())
        scale = np.maximum(1.0, np.abs(ref["p"][hit]).max(axis=1))
        assert (np.abs(gpu["p"][hit] - ref["p"][hit]).max(axis=1) / scale).max() <= 1e-9, what
        assert np.abs(gpu["n"][hit] - ref["n"][hit]).max() <= 1e-9, what
        # the kernels perform the oracle's IEEE operations in the oracle's order: records agree to the bit
        bits = ((gpu["t"].view(np.uint64) == ref["t"].view(np.uint64)) & (gpu["p"].view(np.uint64) == ref["p"].view(np.uint64)).all(axis=1)
                & (gpu["n"].view(np.uint64) == ref["n"].view(np.uint64)).all(axis=1))[hit]
        assert bits.mean() >= 0.9999, "%s: %.6f of hit records bit-identical" % (what, bits.mean())
    return frac


RECIPES = {
    "scene1": (scenes.scene1, dict(width=200, height=112)),
    "scene2": (scenes.scene2, dict(width=212, height=120)),
    "scene3": (scenes.scene3, dict(width=320, height=180)),
    "mesh": (S.small_mesh_scene, dict()),
}


@pytest.mark.parametrize("name", ["scene1", "scene2", "scene3", "mesh"])
@pytest.mark.parametrize("rule", [O.HIT_FIRST_INDEX, O.HIT_NEAREST])
def test_intersect_primary_rays(ctx, name, rule):
    recipe, kw = RECIPES[name]
    osc, ocam, _ = S.oracle_scene(recipe, hit_rule=rule, **kw)
    gsc, _, _ = S.library_scene(ctx, recipe, hit_rule=rule, **kw)
    rays = S.primary_rays(ocam)
    check_hits(gsc.intersect_batch(rays), osc.intersect_batch(rays), "%s primary rule %d" % (name, rule))


@pytest.mark.parametrize("name", ["scene1", "scene2", "mesh"])
@pytest.mark.parametrize("rule", [O.HIT_FIRST_INDEX, O.HIT_NEAREST])
def test_intersect_random_rays(ctx, name, rule):
    recipe, kw = RECIPES[name]
    osc, _, _ = S.oracle_scene(recipe, hit_rule=rule, **kw)
    gsc, _, _ = S.library_scene(ctx, recipe, hit_rule=rule, **kw)
    rays = S.random_rays(1 << 16, seed=11, center=(0.0, 0.0, 0.5), radius=2.8 if name == "scene1" else 6.0, spread=1.5)
    gpu, st = gsc.intersect_batch(rays, want_stats=True)
    check_hits(gpu, osc.intersect_batch(rays), "%s random rule %d" % (name, rule))
    assert st["segments"] == len(rays)
    if name != "scene3":
        assert st["nodes_visited"] > 0 and st["prims_tested"] > 0


def test_intersect_edge_cases(ctx):
    from raytracing_amd import api
    from raytracing_amd.scene_builder import SceneBuilder
    # empty scene: every ray misses
    empty = api.Scene(ctx, *SceneBuilder().arrays())
    rays = S.random_rays(1000, seed=3)
    hits = empty.intersect_batch(rays)
    assert (hits["obj"] == O.MISS).all() and (hits["prim"] == O.MISS).all()
    # zero rays is a valid call
    assert len(empty.intersect_batch(rays[:0])) == 0
    # an empty mesh and a one-triangle mesh
    b = SceneBuilder()
    b.add_model(np.zeros((0, 9)))
    b.add_model(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], dtype=np.float64))
    ob = SceneBuilder(O.from_rgb)
    ob.add_model(np.zeros((0, 9)))
    ob.add_model(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], dtype=np.float64))
    one = np.zeros(2, dtype=O.RAY_DT)
    one["o"] = [[0.0, 0.0, -1.0], [2.0, 2.0, -1.0]]
    one["d"] = [[0.0, 0.0, 1.0], [0.0, 0.0, 1.0]]
    g = api.Scene(ctx, *b.arrays()).intersect_batch(one)
    r = O.Scene(*ob.arrays()).intersect_batch(one)
    check_hits(g, r, "one-triangle mesh")
    # the reference's vertex hit (src/shapes/triangle.rs:110): t = 1, p = origin, n = +z
    assert g["obj"][0] == 1 and abs(g["t"][0] - 1.0) < 1e-12 and g["obj"][1] == O.MISS


@pytest.mark.parametrize("name", ["scene1", "scene2", "scene3"])
def test_trace_batch_spectra(ctx, name):
    """Per-path 60-bin radiance at a fixed Philox seed (RandomWalk, depth 10)."""
    recipe, kw = RECIPES[name]
    osc, ocam, _ = S.oracle_scene(recipe, **kw)
    gsc, _, _ = S.library_scene(ctx, recipe, **kw)
    rays = S.primary_rays(ocam)
    ref, ref_segments = osc.trace_batch(rays, O.INT_RANDOM_WALK, 10, seed=5, pixel_base=17, sample_index=3)
    gpu, st = gsc.trace_batch(rays, O.INT_RANDOM_WALK, 10, seed=5, pixel_base=17, sample_index=3)
    identical = (gpu.view(np.uint32) == ref.view(np.uint32)).all(axis=1)
    # The spectra are products of the same table rows in the same order, so a path either matches
    # bit for bit or took a different turn at a geometric near-tie; allow 1 path in 10 000 to do so.
    assert identical.mean() >= 0.9999, "%s: %.6f of paths bit-identical" % (name, identical.mean())
    assert abs(float(gpu.sum()) - float(ref.sum())) <= 1e-3 * max(1.0, float(ref.sum()))
    assert abs(st["segments"] - ref_segments) <= 1e-4 * ref_segments + 2


@pytest.mark.parametrize("integrator", [O.INT_ALBEDO, O.INT_NORMAL])
def test_trace_batch_aov(ctx, integrator):
    osc, ocam, _ = S.oracle_scene(scenes.scene1, width=200, height=112)
    gsc, _, _ = S.library_scene(ctx, scenes.scene1, width=200, height=112)
    rays = S.primary_rays(ocam)
    ref, _ = osc.trace_batch(rays, integrator, 0, seed=1)
    gpu, _ = gsc.trace_batch(rays, integrator, 0, seed=1)
    if integrator == O.INT_ALBEDO:
        assert (gpu.view(np.uint32) == ref.view(np.uint32)).all()
    else:
        # from_RGB of the normal: a last-bit difference in n moves a bin by a few ulp
        assert np.abs(gpu - ref).max() <= 1e-5


@pytest.mark.parametrize("name,integrator,depth,spp", [
    ("scene1", O.INT_RANDOM_WALK, 10, 8), ("scene2", O.INT_RANDOM_WALK, 10, 4), ("scene3", O.INT_RANDOM_WALK, 10, 8),
    ("scene1", O.INT_ALBEDO, 0, 4), ("scene1", O.INT_NORMAL, 0, 4), ("scene3", O.INT_RANDOM_WALK, 32, 2),
    ("mesh", O.INT_RANDOM_WALK, 10, 4), ("mesh", O.INT_NORMAL, 0, 2),
])
def test_render_matches_oracle(ctx, name, integrator, depth, spp):
    """Whole images at matched spp and seed: 8-bit output and linear mean XYZ."""
    from raytracing_amd import api
    recipe, kw = RECIPES[name]
    osc, ocam, _ = S.oracle_scene(recipe, **kw)
    gsc, gcam, _ = S.library_scene(ctx, recipe, **kw)
    ref8, ref_xyz, ref_lin, ref_st = osc.render(ocam, integrator, depth, spp, seed=9)
    gpu8, gpu_xyz, st = gsc.render(gcam, api.render_params(integrator, depth, spp, seed=9))
    # linear image: per-pixel sums of the same per-sample XYZ values in the same order
    scale = max(1e-6, float(np.abs(ref_lin).mean()))
    rel_mean = abs(float(gpu_xyz.mean()) - float(ref_lin.mean())) / scale
    assert rel_mean <= 1e-4, "relative mean XYZ error %.3e" % rel_mean
    exact = (gpu_xyz == ref_lin).all(axis=2).mean()
    assert exact >= 0.999, "%.5f of pixels have bit-identical mean XYZ" % exact
    # 8-bit image against the reference-order pipeline (sum spectra, then XYZ): RMSE in LSB
    diff = gpu8.astype(np.int32) - ref8.astype(np.int32)
    rmse = float(np.sqrt((diff.astype(np.float64) ** 2).mean()))
    assert rmse <= 0.25, "8-bit RMSE %.4f LSB" % rmse
    assert (diff == 0).mean() >= 0.995
    assert st["paths"] == ref_st.paths
    assert abs(st["segments"] - ref_st.segments) <= 1e-4 * ref_st.segments + 2


def test_partitions_reproduce_full_render(ctx):
    """Sample-range halves and interleaved tile halves, accumulated into one buffer, are bit-identical
    to the single-call render (this is what the multi-GPU ranks do before their one collective)."""
    import torch
    from raytracing_amd import api
    gsc, gcam, _ = S.library_scene(ctx, scenes.scene1, width=200, height=112)
    w, h, spp = gcam.width, gcam.height, 6
    stream = torch.cuda.current_stream().cuda_stream

    def run(parts):
        acc = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
        for p in parts:
            gsc.render_accumulate(gcam, p, acc.data_ptr(), stream)
        torch.cuda.synchronize()
        return acc.cpu().numpy()

    full = run([api.render_params(O.INT_RANDOM_WALK, 10, spp, seed=2)])
    halves = run([api.render_params(O.INT_RANDOM_WALK, 10, spp, seed=2, sample_range=(0, 3)),
                  api.render_params(O.INT_RANDOM_WALK, 10, spp, seed=2, sample_range=(3, 6))])
    tiles = run([api.render_params(O.INT_RANDOM_WALK, 10, spp, seed=2, tile_rank=r, tile_world=3) for r in range(3)])
    assert full.any()
    assert (full.view(np.uint32) == halves.view(np.uint32)).all()
    assert (full.view(np.uint32) == tiles.view(np.uint32)).all()
    # tone mapping on the device equals the host-side render() output
    rgb8 = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda:0")
    acc = torch.from_numpy(full).to("cuda:0")
    ctx.tonemap(acc.data_ptr(), spp, w, h, rgb8.data_ptr(), stream)
    torch.cuda.synchronize()
    direct, _, _ = gsc.render(gcam, api.render_params(O.INT_RANDOM_WALK, 10, spp, seed=2))
    assert (rgb8.cpu().numpy().reshape(h, w, 3) == direct).all()


@pytest.mark.parametrize("rule", [O.HIT_FIRST_INDEX, O.HIT_NEAREST])
@pytest.mark.parametrize("integrator,depth,spp", [(O.INT_RANDOM_WALK, 10, 6), (O.INT_RANDOM_WALK, 20, 2), (O.INT_RANDOM_WALK, 1, 2),
                                                  (O.INT_ALBEDO, 0, 2), (O.INT_NORMAL, 0, 2)])
def test_two_kernel_pipeline_equals_fused_and_oracle(ctx, monkeypatch, rule, integrator, depth, spp):
    """Scenes with meshes run as shade + traverse kernels; RTMI_PIPELINE=fused (read when a scene is
    created) forces the one-kernel-per-bounce form. Both must produce the same bits, and the oracle's."""
    from raytracing_amd import api
    osc, ocam, _ = S.oracle_scene(S.multi_mesh_scene, hit_rule=rule)
    ref8, _, ref_lin, ref_st = osc.render(ocam, integrator, depth, spp, seed=6)
    p = api.render_params(integrator, depth, spp, seed=6)
    monkeypatch.delenv("RTMI_PIPELINE", raising=False)
    two, cam, _ = S.library_scene(ctx, S.multi_mesh_scene, hit_rule=rule)
    monkeypatch.setenv("RTMI_PIPELINE", "fused")
    fused, _, _ = S.library_scene(ctx, S.multi_mesh_scene, hit_rule=rule)
    i2, x2, s2 = two.render(cam, p)
    i1, x1, s1 = fused.render(cam, p)
    assert s2["launches"] > s1["launches"]          # the two scenes really took different routes
    assert (x1.view(np.uint32) == x2.view(np.uint32)).all() and (i1 == i2).all()
    assert s1["segments"] == s2["segments"] and s1["paths"] == s2["paths"] == ref_st.paths
    assert abs(s2["segments"] - ref_st.segments) <= 1e-4 * ref_st.segments + 2
    assert (x2 == ref_lin).all(axis=2).mean() >= 0.999
    assert (i2 == ref8).mean() >= 0.995


def test_multi_mesh_hits(ctx):
    """Closest hit over an object list that interleaves meshes, spheres and lone triangles."""
    for rule in (O.HIT_FIRST_INDEX, O.HIT_NEAREST):
        osc, ocam, _ = S.oracle_scene(S.multi_mesh_scene, hit_rule=rule)
        gsc, _, _ = S.library_scene(ctx, S.multi_mesh_scene, hit_rule=rule)
        for rays in (S.primary_rays(ocam), S.random_rays(20000, 11, center=(0.5, 0.0, 1.0), radius=5.0)):
            check_hits(gsc.intersect_batch(rays), osc.intersect_batch(rays), "multi mesh rule %d" % rule)


def test_degenerate_meshes_through_the_two_kernel_pipeline(ctx):
    """An empty mesh, a one-triangle mesh and a mesh that is the only light, with no sphere or lone triangle at
    all in the second scene: the list / traverse / shade hand-over has nothing to fold with."""
    from raytracing_amd import api
    from raytracing_amd.scene_builder import SceneBuilder, load_golden_mesh

    def with_spheres(from_rgb):
        b = SceneBuilder(from_rgb)
        b.add_model(np.zeros((0, 9)), b.material("Lambertian", (0.5, 0.5, 0.5)))
        b.add_model(np.array([[0.0, -1.0, 0.2, 0.0, 1.0, 0.2, 0.0, 0.0, 1.8]]), b.material("Specular", (0.9, 0.9, 0.9)))
        b.add_sphere((0.0, 0.0, 4.0), 1.5, material=b.material("Emitter", (1.0, 1.0, 1.0)))
        b.add_sphere((0.0, 0.0, -1000.0), 1000.0, material=b.material("Lambertian", (0.4, 0.4, 0.4)))
        return b

    def meshes_only(from_rgb):
        b = SceneBuilder(from_rgb)
        room = load_golden_mesh("cubo")
        b.add_model(room, b.material("Lambertian", (0.6, 0.6, 0.6)))                        # camera inside the cube
        lamp = load_golden_mesh("cubo") * 0.15 + np.tile(np.array([0.0, 0.0, 2.0]), 3)
        b.add_model(lamp, b.material("Emitter", (1.0, 0.9, 0.8)))
        return b

    cam_args = ((-2.5, 0.0, 1.0), 0.1, 100.0, (0.0, 0.2, 0.0), 128, 72)
    for build in (with_spheres, meshes_only):
        osc, ocam = O.Scene(*build(O.from_rgb).arrays()), O.camera(*cam_args)
        gsc, gcam = api.Scene(ctx, *build(None).arrays()), api.camera_new(*cam_args)
        for integrator, depth in ((O.INT_RANDOM_WALK, 6), (O.INT_NORMAL, 0)):
            ref8, _, ref_lin, ref_st = osc.render(ocam, integrator, depth, 4, seed=8)
            gpu8, gpu_xyz, st = gsc.render(gcam, api.render_params(integrator, depth, 4, seed=8))
            assert ref8.any(), build.__name__
            assert st["segments"] == ref_st.segments and st["paths"] == ref_st.paths
            assert (gpu_xyz == ref_lin).all(axis=2).mean() >= 0.999, build.__name__
            assert (gpu8 == ref8).mean() >= 0.995, build.__name__


@pytest.mark.parametrize("rule", [O.HIT_FIRST_INDEX, O.HIT_NEAREST])
def test_rays_with_zero_direction_components(ctx, rule):
    """Rays that run parallel to one or two coordinate axes (direction components exactly zero, of either sign of zero),
    from origins off the mesh's centre planes, through a mesh deep enough to have a real tree: a reciprocal direction of
    +-inf must not turn a node box the ray runs inside of into a miss. Every ray must agree with the oracle's scan."""
    osc, _, _ = S.oracle_scene(S.small_mesh_scene, hit_rule=rule)
    gsc, _, _ = S.library_scene(ctx, S.small_mesh_scene, hit_rule=rule)
    rays = []
    grid = np.linspace(-1.1, 1.1, 23)
    for axis in range(3):
        u, v = [a for a in range(3) if a != axis]
        for a in grid:
            for b in grid:
                for sign in (1.0, -1.0):
                    o = np.array([0.0, 0.0, 1.0])              # the mesh sits around (0, 0, 1)
                    o[u] += a; o[v] += b; o[axis] -= 5.0 * sign
                    d = np.zeros(3); d[axis] = sign
                    rays.append((o, d))
                    d2 = d.copy(); d2[u] = -0.0                # a negative zero
                    rays.append((o, d2))
                    d3 = d.copy(); d3[u] = 0.3                 # one zero component only
                    rays.append((o, d3 / np.linalg.norm(d3)))
    arr = np.zeros(len(rays), dtype=O.RAY_DT)
    arr["o"] = [r[0] for r in rays]
    arr["d"] = [r[1] for r in rays]
    gpu, ref = gsc.intersect_batch(arr), osc.intersect_batch(arr)
    same = (gpu["obj"] == ref["obj"]) & (gpu["prim"] == ref["prim"])
    assert same.all(), "rule %d: %d of %d axis-parallel rays differ, first at %d" % (rule, (~same).sum(), len(arr), int(np.argmax(~same)))
    hit = ref["obj"] != O.MISS
    assert hit.mean() > 0.2
    assert (gpu["t"][hit].view(np.uint64) == ref["t"][hit].view(np.uint64)).all()
    # the same rays through the two-kernel pipeline (traverse_kernel sets its rays up separately)
    spectra_g, _ = gsc.trace_batch(arr, O.INT_NORMAL, 0, seed=0)
    spectra_r, _ = osc.trace_batch(arr, O.INT_NORMAL, 0, seed=0)
    assert np.abs(spectra_g - spectra_r).max() <= 1e-5


def test_model_box_gate_on_faces_and_edges(ctx):
    """ModelObj::intersect first tests the ray against the model's own box (grown from the origin point,
    src/shapes/model_obj.rs:62-66,78) with exact binary64 slabs (src/geometry/aabb.rs:136-247); a ray that fails it
    reports no hit even where Moller-Trumbore accepts a triangle lying in the box's surface. Rays in the faces and
    along the edges of axis-aligned cubes, from origins on, just inside and just outside the box planes, must come out
    as the oracle's linear scan behind the same gate reports them -- every one of them, not 99.99 %."""
    from raytracing_amd import api
    from raytracing_amd.scene_builder import SceneBuilder, load_golden_mesh

    def build(from_rgb):
        b = SceneBuilder(from_rgb)
        b.add_model(load_golden_mesh("cubo") * 0.5 + np.tile(np.array([1.5, 0.25, 0.75]), 3), b.material("Lambertian", (0.5, 0.5, 0.5)))
        b.add_model(load_golden_mesh("cubo"), b.material("Lambertian", (0.6, 0.6, 0.6)))   # +-3: contains the origin
        return b

    dirs = [d for d in np.array(np.meshgrid([-1, 0, 1], [-1, 0, 1], [-1, 0, 1])).T.reshape(-1, 3) if d.any()]
    dirs += [(1.0, 1e-17, 0.0), (0.0, 1.0, 1e-300), (1e-9, 1.0, -1e-9), (0.3, -0.7, 0.2)]
    dirs = np.array(dirs, dtype=np.float64)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    rays = []
    for lo, hi in (((0.0, -1.25, -0.75), (3.0, 1.75, 2.25)), ((-3.0, -3.0, -3.0), (3.0, 3.0, 3.0)), ((0.0, 0.0, 0.0), (3.0, 1.75, 2.25))):
        per_axis = []
        for k in range(3):
            l, h = lo[k], hi[k]
            per_axis.append([np.nextafter(l, -np.inf), l, np.nextafter(l, np.inf), 0.5 * (l + h), np.nextafter(h, -np.inf), h,
                             np.nextafter(h, np.inf), l - 2.0, h + 2.0])
        for x in per_axis[0]:
            for y in per_axis[1]:
                for z in per_axis[2]:
                    for d in dirs:
                        rays.append(((x, y, z), d))
    arr = np.zeros(len(rays), dtype=O.RAY_DT)
    arr["o"] = [r[0] for r in rays]
    arr["d"] = [r[1] for r in rays]
    for rule in (O.HIT_FIRST_INDEX, O.HIT_NEAREST):
        osc = O.Scene(*build(O.from_rgb).arrays(), hit_rule=rule)
        gsc = api.Scene(ctx, *build(None).arrays(), hit_rule=rule)
        gpu, ref = gsc.intersect_batch(arr), osc.intersect_batch(arr)
        same = (gpu["obj"] == ref["obj"]) & (gpu["prim"] == ref["prim"])
        assert same.all(), "rule %d: %d of %d edge rays differ, first at %d" % (rule, (~same).sum(), len(arr), int(np.argmax(~same)))
        hit = ref["obj"] != O.MISS
        assert hit.any() and (~hit).any()
        assert (gpu["t"][hit].view(np.uint64) == ref["t"][hit].view(np.uint64)).all()
        # the same rays as level-0 rays of the two-kernel pipeline (traverse_kernel applies the gate there)
        spectra_g, _ = gsc.trace_batch(arr, O.INT_ALBEDO, 0, seed=0)
        spectra_r, _ = osc.trace_batch(arr, O.INT_ALBEDO, 0, seed=0)
        assert (spectra_g.view(np.uint32) == spectra_r.view(np.uint32)).all()


def test_sealed_scene_fills_every_queue_region(ctx):
    """Inside a closed Lambertian box whose faces look inward every path survives every bounce, so each queue level holds
    the whole batch and every region of the queues receives as many paths as it can. Batch sizes that are multiples
    of 256 * 64 (and some that are not) must neither drop a path nor change a bit of the image."""
    from raytracing_amd import api
    from raytracing_amd.scene_builder import SceneBuilder, load_golden_mesh

    def build(from_rgb):
        b = SceneBuilder(from_rgb)
        inward = load_golden_mesh("cubo").reshape(-1, 3, 3)[:, ::-1, :].reshape(-1, 9)    # reversed winding
        b.add_model(inward, b.material("Lambertian", (0.9, 0.9, 0.9)))
        lamp = load_golden_mesh("cubo") * 0.01 + np.tile(np.array([0.0, 0.0, 2.5]), 3)
        b.add_model(lamp, b.material("Emitter", (1.0, 1.0, 1.0)))
        return b

    cam_args = ((0.5, 0.25, 0.0), 0.1, 110.0, (0.0, 0.1, 0.3), 192, 128)
    osc, ocam = O.Scene(*build(O.from_rgb).arrays()), O.camera(*cam_args)
    gcam = api.camera_new(*cam_args)
    p = api.render_params(O.INT_RANDOM_WALK, 12, 6, seed=21)
    ref8, _, ref_lin, ref_st = osc.render(ocam, O.INT_RANDOM_WALK, 12, 6, seed=21)
    assert ref_st.segments >= 0.98 * 12 * ref_st.paths       # nearly every path runs to the depth limit
    images = []
    for batch in (0, 16384, 3 * 16384, 8192, 1 << 18):
        c = ctx if batch == 0 else api.Context(0, max_batch_paths=batch)
        gsc = api.Scene(c, *build(None).arrays())
        img, xyz, st = gsc.render(gcam, p)                    # raises if a queue region overflowed
        assert st["segments"] == ref_st.segments and st["paths"] == ref_st.paths, batch
        images.append(xyz)
        gsc.close()
        if batch:
            c.close()
    for xyz in images[1:]:
        assert (xyz.view(np.uint32) == images[0].view(np.uint32)).all()
    assert (images[0] == ref_lin).all(axis=2).mean() >= 0.999


def test_small_batches_equal_large(ctx):
    """The batch size only changes scheduling, never the image."""
    from raytracing_amd import api
    small = api.Context(0, max_batch_paths=4096)
    for recipe in (scenes.scene3, scenes.scene1):      # without meshes (fused kernel) and with one (two-kernel pipeline)
        a, acam, _ = S.library_scene(ctx, recipe, width=160, height=90)
        b, bcam, _ = S.library_scene(small, recipe, width=160, height=90)
        p = api.render_params(O.INT_RANDOM_WALK, 10, 5, seed=4)
        ia, xa, sa = a.render(acam, p)
        ib, xb, sb = b.render(bcam, p)
        assert (ia == ib).all() and (xa.view(np.uint32) == xb.view(np.uint32)).all()
        assert sa["segments"] == sb["segments"] and sb["batches"] > sa["batches"]
    small.close()


def test_depth_accounting_and_black_materials(ctx):
    """depth 0 and 1, a default (colourless emitter) material and a Lambertian without ambient colour."""
    from raytracing_amd import api
    from raytracing_amd.scene_builder import SceneBuilder

    def build(from_rgb):
        b = SceneBuilder(from_rgb)
        b.add_sphere((0.0, 0.0, 1.0), 1.0)                                       # Material::default()
        b.add_sphere((0.0, 2.5, 1.0), 1.0, material=b._material(O.MAT_LAMBERTIAN))  # no ambient colour
        b.add_sphere((0.0, -2.5, 1.0), 1.0, material=b.material("Emitter", (0.9, 0.8, 0.7)))
        b.add_sphere((0.0, 0.0, -1000.0), 1000.0, material=b.material("Lambertian", (0.5, 0.5, 0.5)))
        return b
    cam_args = ((-6.0, 0.0, 2.0), 0.1, 90.0, (0.0, 0.3, 0.0), 128, 72)
    osc, ocam = O.Scene(*build(O.from_rgb).arrays()), O.camera(*cam_args)
    gsc, gcam = api.Scene(ctx, *build(None).arrays()), api.camera_new(*cam_args)
    for depth in (0, 1, 2, 5):
        ref8, _, ref_lin, ref_st = osc.render(ocam, O.INT_RANDOM_WALK, depth, 4, seed=1)
        gpu8, gpu_xyz, st = gsc.render(gcam, api.render_params(O.INT_RANDOM_WALK, depth, 4, seed=1))
        assert st["segments"] == ref_st.segments, depth
        assert (gpu_xyz == ref_lin).all(axis=2).mean() >= 0.999
        assert (gpu8 == ref8).mean() >= 0.995
        if depth == 0:
            assert not gpu8.any()


def test_benchmark_size_properties(ctx):
    """The benchmark configuration itself (Suzanne, 1920x1080, 64 spp, depth 10: 133 M paths, 267 M segments), where
    the oracle would need minutes: properties that do not depend on the size. A render is reproducible to the bit;
    sample ranges and interleaved tiles accumulate to the bits of the single call; segments add up; a context with
    small batches produces the same image; another seed gives another estimate of the same mean luminance. (The
    image itself is checked against the oracle at small sizes above: the mean depends on the resolution, because the
    jittered pixel footprints are shifted by half a pixel, src/integrators/integrator.rs:130-135.)"""
    import torch
    from raytracing_amd import api
    w, h, spp, depth = 1920, 1080, 64, 10
    gsc, gcam, _ = S.library_scene(ctx, scenes.scene2, width=w, height=h)
    stream = torch.cuda.current_stream().cuda_stream

    def run(scene, parts):
        acc = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
        segments = paths = 0
        for p in parts:
            scene.render_accumulate(gcam, p, acc.data_ptr(), stream)
            st = scene.collect_stats(stream)
            segments += st["segments"]; paths += st["paths"]
        torch.cuda.synchronize()
        return acc, segments, paths

    P = lambda **kw: api.render_params(O.INT_RANDOM_WALK, depth, spp, seed=0, **kw)
    full, seg_full, paths_full = run(gsc, [P()])
    again, seg_again, _ = run(gsc, [P()])
    assert paths_full == w * h * spp and paths_full < seg_full <= depth * paths_full
    assert torch.equal(full.view(torch.int32), again.view(torch.int32)) and seg_full == seg_again
    quarters, seg_q, paths_q = run(gsc, [P(sample_range=(a, a + 16)) for a in range(0, spp, 16)])
    assert torch.equal(full.view(torch.int32), quarters.view(torch.int32)) and (seg_q, paths_q) == (seg_full, paths_full)
    tiles, seg_t, paths_t = run(gsc, [P(tile_rank=r, tile_world=2) for r in range(2)])
    assert torch.equal(full.view(torch.int32), tiles.view(torch.int32)) and (seg_t, paths_t) == (seg_full, paths_full)
    small = api.Context(0, max_batch_paths=1 << 22)
    ssc, _, _ = S.library_scene(small, scenes.scene2, width=w, height=h)
    batched, seg_b, _ = run(ssc, [P()])
    assert torch.equal(full.view(torch.int32), batched.view(torch.int32)) and seg_b == seg_full
    small.close()
    # the one-kernel-per-bounce form of the pipeline (no per-pixel culling, no traversal lists) gives the same bits
    import os
    os.environ["RTMI_PIPELINE"] = "fused"
    try:
        fsc, _, _ = S.library_scene(ctx, scenes.scene2, width=w, height=h)
    finally:
        del os.environ["RTMI_PIPELINE"]
    p8 = api.render_params(O.INT_RANDOM_WALK, depth, 8, seed=0)
    a8, seg_a, _ = run(gsc, [p8])
    f8, seg_f, _ = run(fsc, [p8])
    assert torch.equal(a8.view(torch.int32), f8.view(torch.int32)) and seg_a == seg_f
    # another seed is another Monte-Carlo estimate of the same image: the mean luminance agrees within its noise
    other, _, _ = run(gsc, [api.render_params(O.INT_RANDOM_WALK, depth, spp, seed=77)])
    mean_y, mean_other = float(full.view(h, w, 3)[:, :, 1].mean()), float(other.view(h, w, 3)[:, :, 1].mean())
    assert mean_y > 0.0 and abs(mean_y - mean_other) <= 0.005 * mean_y and not torch.equal(full, other)
    assert torch.isfinite(full).all() and float(full.min()) >= 0.0


def test_bad_arguments(ctx):
    from raytracing_amd import api, _lib as L
    gsc, gcam, _ = S.library_scene(ctx, scenes.scene3, width=64, height=36)
    with pytest.raises(L.RtmiError) as e:
        gsc.render(gcam, api.render_params(O.INT_RANDOM_WALK, 33, 1))
    assert e.value.code == L.UNSUPPORTED
    with pytest.raises(L.RtmiError) as e:
        gsc.render(gcam, api.render_params(7, 1, 1))
    assert e.value.code == L.BAD_ARG
    with pytest.raises(L.RtmiError) as e:
        gsc.render(gcam, api.render_params(O.INT_RANDOM_WALK, 1, 0))
    assert e.value.code == L.BAD_ARG
    from raytracing_amd.scene_builder import SceneBuilder
    b = SceneBuilder()
    b.add_sphere((0, 0, 0), 1.0, scale=(1.0, 0.0, 1.0))  # singular transform
    with pytest.raises(L.RtmiError) as e:
        api.Scene(ctx, *b.arrays())
    assert e.value.code == L.BAD_ARG
