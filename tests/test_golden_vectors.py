"""The committed golden vectors (tests/golden/hits.npz, made by tests/golden/make_golden_hits.py):
the oracle must still reproduce them bit for bit (CPU), and the HIP library is checked against them
without the oracle in the loop (GPU)."""
from pathlib import Path

import numpy as np
import pytest

import _oracle as O
import _scenes as S
from raytracing_amd import scenes

GOLDEN = Path(__file__).resolve().parent / "golden" / "hits.npz"
RECIPES = {"scene1": scenes.scene1, "scene2": scenes.scene2, "scene3": scenes.scene3, "mesh": S.small_mesh_scene}
KW = dict(width=48, height=27)


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.mark.parametrize("name", sorted(RECIPES))
def test_oracle_reproduces_golden(golden, name):
    rays = golden[name + "_rays"]
    for rule, tag in ((O.HIT_FIRST_INDEX, "first"), (O.HIT_NEAREST, "nearest")):
        sc, _, _ = S.oracle_scene(RECIPES[name], hit_rule=rule, **KW)
        hits = sc.intersect_batch(rays)
        assert hits.tobytes() == golden["%s_hits_%s" % (name, tag)].tobytes(), (name, tag)
    sc, _, _ = S.oracle_scene(RECIPES[name], **KW)
    spectra, segments = sc.trace_batch(rays[:48 * 27], O.INT_RANDOM_WALK, 10, seed=5)
    assert segments == int(golden[name + "_segments"][0])
    assert spectra.tobytes() == golden[name + "_spectra"].tobytes()


def test_golden_rules_differ_only_inside_meshes(golden):
    """First-index and nearest agree wherever a ray meets at most one triangle of a mesh."""
    a, b = golden["scene3_hits_first"], golden["scene3_hits_nearest"]
    assert a.tobytes() == b.tobytes()                      # no mesh in scene3
    a, b = golden["scene2_hits_first"], golden["scene2_hits_nearest"]
    differ = (a["prim"] != b["prim"]) | (a["obj"] != b["obj"])
    assert differ.any() and (a["obj"][differ] == 0).any()   # Suzanne is object 0
    same_prim = ~differ & (a["obj"] == 0)
    assert (a["t"][same_prim] == b["t"][same_prim]).all()
    assert (b["t"][differ & (b["obj"] == 0) & (a["obj"] == 0)] <= a["t"][differ & (b["obj"] == 0) & (a["obj"] == 0)]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(RECIPES))
def test_library_matches_golden(golden, name):
    from raytracing_amd import api
    ctx = api.Context(0)
    rays = golden[name + "_rays"]
    for rule, tag in ((O.HIT_FIRST_INDEX, "first"), (O.HIT_NEAREST, "nearest")):
        sc, _, _ = S.library_scene(ctx, RECIPES[name], hit_rule=rule, **KW)
        hits = sc.intersect_batch(rays)
        ref = golden["%s_hits_%s" % (name, tag)]
        same = (hits["obj"] == ref["obj"]) & (hits["prim"] == ref["prim"])
        assert same.mean() >= 0.9999, (name, tag, same.mean())
        hit = same & (ref["obj"] != O.MISS)
        rel = np.abs(hits["t"][hit] - ref["t"][hit]) / np.maximum(np.abs(ref["t"][hit]), 1e-300)
        assert rel.max() <= 1e-5
    sc, _, _ = S.library_scene(ctx, RECIPES[name], **KW)
    spectra, st = sc.trace_batch(rays[:48 * 27], O.INT_RANDOM_WALK, 10, seed=5)
    ref = golden[name + "_spectra"]
    identical = (spectra.view(np.uint32) == ref.view(np.uint32)).all(axis=1)
    assert identical.mean() >= 0.9999, identical.mean()
    assert abs(st["segments"] - int(golden[name + "_segments"][0])) <= 3
    ctx.close()
