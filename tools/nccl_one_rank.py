"""The collectives bench.py issues when it runs on several GPUs, on a one-rank RCCL group (API and dtype check)."""
import os, sys, time
from pathlib import Path
sys.path[:0] = [str(Path(__file__).resolve().parent.parent)]
import torch, torch.distributed as dist
from raytracing_amd import parallel
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
print("init ok", flush=True)
from raytracing_amd import api, scenes, _lib as L
ctx = api.Context(0)
b, cam_args = scenes.scene2(width=480, height=270)
sc = api.Scene(ctx, *b.arrays()); cam = api.camera_new(*cam_args)
acc = torch.zeros(480 * 270 * 3, dtype=torch.float32, device=device)
stream = torch.cuda.current_stream().cuda_stream
sc.render_accumulate(cam, api.render_params(L.INT_RANDOM_WALK, 10, 8, seed=0, **parallel.rank_share("samples", 8, 0, 1)), acc.data_ptr(), stream)
parallel.merge_sums(acc, 0); print("render + reduce ok", float(acc.sum()), flush=True)
t = torch.tensor([1.5, 2.0], dtype=torch.float64, device=device)
m = t.clone(); dist.all_reduce(m, op=dist.ReduceOp.MAX); dist.all_reduce(t, op=dist.ReduceOp.SUM)
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
print("all_reduce + barrier ok", m.tolist(), t.tolist(), flush=True)
dist.destroy_process_group()
print("done", flush=True)
