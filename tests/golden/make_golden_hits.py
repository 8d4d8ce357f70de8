#!/usr/bin/env python3
"""Generates tests/golden/hits.npz: fixed ray sets with the CPU oracle's answers.

Per workload (the three stock scenes and a small procedural mesh): pixel-centre primary rays of a
48x27 camera plus 2048 seeded random rays; the oracle's hit records under both hit rules; and the
oracle's 60-bin radiance for the primary rays (RandomWalk, depth 10, Philox seed 5). The fixture
pins the oracle against drift (CPU test) and is a second, oracle-free reference for the GPU test.
Run from the repository root: python tests/golden/make_golden_hits.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import _oracle as O  # noqa: E402
import _scenes as S  # noqa: E402
from raytracing_amd import scenes  # noqa: E402

WORKLOADS = {
    "scene1": (scenes.scene1, dict(width=48, height=27), dict(center=(0.0, 0.0, 0.5), radius=2.8, spread=1.5)),
    "scene2": (scenes.scene2, dict(width=48, height=27), dict(center=(0.0, 0.0, 0.5), radius=6.0, spread=1.5)),
    "scene3": (scenes.scene3, dict(width=48, height=27), dict(center=(0.0, 0.0, 0.5), radius=6.0, spread=1.5)),
    "mesh": (S.small_mesh_scene, dict(width=48, height=27), dict(center=(0.0, 0.0, 1.0), radius=6.0, spread=1.2)),
}


def main():
    out = {}
    for name, (recipe, kw, rr) in WORKLOADS.items():
        _, cam, _ = S.oracle_scene(recipe, **kw)
        rays = np.concatenate([S.primary_rays(cam), S.random_rays(2048, seed=23, **rr)])
        out[name + "_rays"] = rays
        for rule, tag in ((O.HIT_FIRST_INDEX, "first"), (O.HIT_NEAREST, "nearest")):
            sc, _, _ = S.oracle_scene(recipe, hit_rule=rule, **kw)
            out["%s_hits_%s" % (name, tag)] = sc.intersect_batch(rays, threads=1)
        sc, _, _ = S.oracle_scene(recipe, **kw)
        spectra, segments = sc.trace_batch(rays[:48 * 27], O.INT_RANDOM_WALK, 10, seed=5, pixel_base=0, sample_index=0, threads=1)
        out[name + "_spectra"] = spectra
        out[name + "_segments"] = np.array([segments], dtype=np.uint64)
        print(name, len(rays), "rays,", int((out[name + "_hits_first"]["obj"] != O.MISS).sum()), "hits,", segments, "segments")
    path = Path(__file__).resolve().parent / "hits.npz"
    np.savez_compressed(path, **out)
    print("wrote", path, path.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
