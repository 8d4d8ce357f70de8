"""bench.py must start when typed as `python bench.py --gpus N` (it spawns its own N ranks) and when launched by
torch.distributed.run; --dry-run exercises launcher, rendezvous and merge over gloo, without a GPU."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_self_launch_two_ranks():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr
    out = _line(r.stdout)
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["merge_ok"]
    shares = [tuple(p["share"]["sample_range"]) for p in sorted(out["per_rank"], key=lambda p: p["rank"])]
    assert shares == [(0, 64), (64, 128)]          # weak scaling in samples: every rank its own 64 of a 128-spp job
    assert out["scaling"] == "weak" and out["spp_job"] == 128


def test_under_torch_distributed_run_with_tiles():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--partition", "tiles"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr
    out = _line(r.stdout)
    assert out["world_size"] == 2 and out["merge_ok"]
    assert [p["share"] for p in sorted(out["per_rank"], key=lambda p: p["rank"])] == [{"tile_rank": 0, "tile_world": 2}, {"tile_rank": 1, "tile_world": 2}]


def test_strong_scaling_splits_the_samples():
    """--scaling strong under the sample partition: the job keeps the workload's 64 spp and every rank renders its share of
    them (the 256-spp-over-8-GPUs form of the million-triangle configuration), where the default gives every rank 64."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--scaling", "strong", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr
    out = _line(r.stdout)
    assert out["world_size"] == 2 and out["merge_ok"] and out["scaling"] == "strong" and out["spp_job"] == 64
    shares = [tuple(p["share"]["sample_range"]) for p in sorted(out["per_rank"], key=lambda p: p["rank"])]
    assert shares == [(0, 32), (32, 64)]


def test_world_size_mismatch_is_an_error():
    env = dict(_env(), WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
