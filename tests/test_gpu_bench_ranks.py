"""The rank path of bench.py with a real GPU under it: two ranks started by bench.py's own launcher, both rendering on
device 0, sums merged over gloo (RCCL refuses two ranks on one device, so the collective itself is the one thing this
rehearsal replaces). Everything else -- shares, per-rank contexts and workspaces side by side on one card, fences, the
max-over-ranks timing, the gathered statistics, rank 0's line -- is what an N-GPU run executes."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "scene1_400x225_100spp", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--sustain", "0.3"] + extra, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_by_samples_and_by_tiles():
    one = _run([])
    two = _run(["--gpus", "2", "--share-gpu"])
    assert two["n_gpus"] == 2 and two["world_size"] == 2 and two["backend"] == "gloo"
    assert len(two["per_rank_ms_per_step"]) == 2 and all(ms > 0 for ms in two["per_rank_ms_per_step"])
    # the sustained leg: the same step for a stated stretch after the timed ones (every rank takes the same number of steps)
    for line in (one, two):
        sus = line["sustained"]
        assert sus["steps"] >= 1 and 0.0 < sus["ms_per_step"] < 50 * line["ms_per_step"] and sus["value"] > 0
    assert two["scaling"] == "weak" and two["config"]["spp_job"] == 2 * one["config"]["spp_job"]
    # weak scaling in samples: the job has twice the paths; its segment count is within a per cent of twice one rank's
    seg1 = one["config"]["segments_per_step_per_gpu"]
    total2 = two["value"] * 1e6 * two["ms_per_step"] * 1e-3
    assert abs(total2 - 2 * seg1) < 0.01 * 2 * seg1
    tiles = _run(["--gpus", "2", "--share-gpu", "--partition", "tiles"])
    assert tiles["scaling"] == "strong" and tiles["config"]["spp_job"] == one["config"]["spp_job"]
    total_t = tiles["value"] * 1e6 * tiles["ms_per_step"] * 1e-3
    assert round(total_t) == seg1          # the same paths, dealt out by tiles: the same segments exactly


def test_strong_scaling_by_samples_and_the_in_process_devices_leg():
    """--scaling strong: two ranks split the workload's samples, so the job's segments are one rank's; --devices N runs
    the same split inside one process through rtmi_multi_render (here two shares on device 0: copies and adds)."""
    one = _run([])
    seg1 = one["config"]["segments_per_step_per_gpu"]
    strong = _run(["--gpus", "2", "--share-gpu", "--scaling", "strong"])
    assert strong["scaling"] == "strong" and strong["config"]["spp_job"] == one["config"]["spp_job"]
    assert strong["config"]["spp_per_gpu"] == one["config"]["spp_job"] / 2
    assert round(strong["value"] * 1e6 * strong["ms_per_step"] * 1e-3) == seg1
    assert one["mesh_rays_per_step"] > 0 and abs(strong["mesh_rays_per_step"] - one["mesh_rays_per_step"]) < 1
    multi = _run(["--devices", "2", "--share-gpu"])
    assert multi["n_gpus"] == 2 and multi["scaling"] == "strong" and multi["backend"].startswith("rtmi_multi")
    assert round(multi["value"] * 1e6 * multi["ms_per_step"] * 1e-3) == seg1 and sum(multi["per_rank_segments"]) >= seg1
    single = _run(["--devices", "1"])
    assert round(single["value"] * 1e6 * single["ms_per_step"] * 1e-3) == seg1
