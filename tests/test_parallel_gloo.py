"""The multi-rank path on CPU: two gloo ranks render their share (the oracle stands in for the GPU
renderer), merge with the same collective the GPU ranks use, and rank 0 checks the result against a
single-process render. Covers both partitions of raytracing_amd/parallel.py."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, kind, result_path):
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    import _oracle as O
    import _scenes as S
    from raytracing_amd import parallel, scenes
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, cam, _ = S.oracle_scene(scenes.scene1, width=96, height=54)
    w, h, spp, depth = cam.width, cam.height, 5, 6

    def accumulate(acc, share):
        buf = acc.numpy().reshape(h, w, 3)
        sc.render(cam, O.INT_RANDOM_WALK, depth, spp, seed=7, sample_range=share.get("sample_range"),
                  tile_rank=share.get("tile_rank", 0), tile_world=share.get("tile_world", 0), xyz_sum=buf, want_lin=False, threads=2)

    def tonemap(acc, n):
        import ctypes as C
        xyz = (acc.numpy().reshape(-1, 3) / np.float32(n)).astype(np.float32)
        out = np.zeros((h * w, 3), dtype=np.uint8)
        rgb = (C.c_float * 3)()
        px = (C.c_uint8 * 3)()
        for i in range(h * w):
            O.lib().ref_xyz_to_rgb(xyz[i].ctypes.data_as(C.POINTER(C.c_float)), rgb)
            O.lib().ref_tonemap(rgb, px)
            out[i] = list(px)
        return out.reshape(h, w, 3), xyz.reshape(h, w, 3)

    img = parallel.render_distributed(accumulate, tonemap, h * w * 3, spp, kind=kind, device="cpu")
    if rank == 0:
        rgb8, xyz = img
        ref8, _, ref_lin, _ = sc.render(cam, O.INT_RANDOM_WALK, depth, spp, seed=7, threads=2)
        np.savez(result_path, same8=(rgb8 == ref8).mean(), exact=(xyz == ref_lin).all(axis=2).mean(),
                 maxrel=np.abs(xyz - ref_lin).max() / max(1e-9, np.abs(ref_lin).max()), nonzero=float(xyz.any()))
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["samples", "tiles"])
def test_two_ranks_reproduce_single_process(tmp_path, kind):
    result = tmp_path / ("result_%s.npz" % kind)
    mp.spawn(_worker, args=(2, _free_port(), kind, str(result)), nprocs=2, join=True)
    r = np.load(result)
    assert r["nonzero"] == 1.0
    if kind == "tiles":
        # disjoint tiles merged by a sum with zeros: bit-identical to the single-process render
        assert r["exact"] == 1.0 and r["same8"] == 1.0
    else:
        # two partial sums added once instead of one long sequential sum: differs only by float rounding
        assert r["maxrel"] <= 1e-6 and r["same8"] >= 0.999


def test_sample_ranges_cover_everything():
    from raytracing_amd import parallel
    for spp in (1, 5, 64, 100):
        for world in (1, 2, 3, 8):
            ranges = [parallel.sample_range(spp, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == spp
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.rank_share("tiles", 8, 3, 8) == {"tile_rank": 3, "tile_world": 8}
    assert parallel.rank_share("samples", 8, 0, 1) == {}
