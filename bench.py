#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X: Mpath-segments/s, whole job.

One "step" = one full render of the workload (every pixel, every sample, depth-10 random walk)
from a scene that is already resident in HBM, including the merge of the per-rank buffers and the
tone mapping. With N > 1 ranks (torch.distributed over RCCL, one process per GPU) every rank
renders its own `spp` samples per pixel of an N*spp job (sample-range partition, weak scaling) and
one reduce merges the per-pixel XYZ sums. Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 3 --warmup 1
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, ~8 TB/s
NODE_BYTES, TRI_BYTES = 64, 80
# Algorithmic bytes per unit of the two kernels a roofline is quoted for:
#   traverse_kernel, per listed ray: list entry (4) + ray origin and direction (48) + the hit record it folds its result
#     into (12 read, 28 written) + the BVH nodes and triangles that ray visits
#   bounce_kernel (scenes without meshes), per segment: one path-state record read and one written (6 f64 + 4 u32 + 16
#     chain bytes each)
TRAVERSE_RAY_BYTES = 4 + 48 + 12 + 28
# flops model of the same units: 40 per node visit, 55 per triangle test, 180 + 45 per analytic object for shading
NODE_FLOPS, TRI_FLOPS, SHADE_FLOPS = 40, 55, 180 + 6 * 45
VALU_F64_PEAK_TFLOPS = 78.6
STATE_BYTES = 80

WORKLOADS = {
    # the headline configuration: Suzanne (968 triangles) + emitter + ground spheres, 1920x1080, 64 spp
    "suzanne_1080p_64spp": dict(recipe="scene2", width=1920, height=1080, spp=64, depth=10),
    "suzanne_specular_1080p_64spp": dict(recipe="scene2", width=1920, height=1080, spp=64, depth=10, specular_mesh=True),
    "scene1_400x225_100spp": dict(recipe="scene1", width=400, height=225, spp=100, depth=10),
    "scene3_720p_100spp": dict(recipe="scene3", width=1280, height=720, spp=100, depth=10),
    "mesh1m_4k_16spp": dict(recipe="synthetic_mesh_scene", width=3840, height=2160, spp=16, depth=10),
    # scene3's layout with glossy / dielectric extension materials, depth 32 (16 of the configuration's 1024 spp)
    "scene3_extended_4k_16spp": dict(recipe="scene3_extended", width=3840, height=2160, spp=16, depth=32),
}


def build_scene(ctx, w, hit_rule, from_rgb=None, library=True):
    from raytracing_amd import api, scenes
    kw = {k: v for k, v in w.items() if k in ("width", "height", "specular_mesh")}
    builder, cam_args = getattr(scenes, w["recipe"])(from_rgb=from_rgb, **kw)
    if library:
        return api.Scene(ctx, *builder.arrays(), hit_rule=hit_rule), api.camera_new(*cam_args)
    return builder, cam_args


def cpu_baseline(w, hit_rule, spp_sample, threads):
    """The CPU oracle (linear scans, the reference's algorithm) on the first `spp_sample` samples per
    pixel of the same scene and camera, on this machine's host cores."""
    import _oracle as O
    builder, cam_args = build_scene(None, w, hit_rule, from_rgb=O.from_rgb, library=False)
    scene = O.Scene(*builder.arrays(), hit_rule=hit_rule)
    cam = O.camera(*cam_args)
    _, _, _, st = scene.render(cam, O.INT_RANDOM_WALK, w["depth"], w["spp"], seed=0, sample_range=(0, spp_sample),
                               threads=threads, want_lin=False)
    return st.segments / st.seconds / 1e6, st.seconds, st.segments


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="suzanne_1080p_64spp", choices=sorted(WORKLOADS))
    ap.add_argument("--rule", default="first_index", choices=["first_index", "nearest"])
    ap.add_argument("--partition", default="samples", choices=["samples", "tiles"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-paths", type=int, default=0, help="path slots per batch of launches (0 = library default)")
    ap.add_argument("--cpu-spp", type=int, default=0, help="samples per pixel of the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracing_amd import api, parallel, _lib as L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" %
                         (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    w = WORKLOADS[args.workload]
    hit_rule = L.HIT_FIRST_INDEX if args.rule == "first_index" else L.HIT_NEAREST
    ctx = api.Context(local_rank, args.batch_paths)
    scene, cam = build_scene(ctx, w, hit_rule)
    width, height, spp, depth = cam.width, cam.height, w["spp"], w["depth"]
    spp_job = spp * world if args.partition == "samples" else spp     # weak scaling in samples
    stream = torch.cuda.current_stream().cuda_stream
    acc = torch.zeros(height * width * 3, dtype=torch.float32, device=device)
    rgb8 = torch.zeros(height * width * 3, dtype=torch.uint8, device=device)
    share = parallel.rank_share(args.partition, spp_job, rank, world)

    def step(flags=0):
        acc.zero_()
        p = api.render_params(L.INT_RANDOM_WALK, depth, spp_job, seed=0, flags=flags, **share)
        scene.render_accumulate(cam, p, acc.data_ptr(), stream)
        parallel.merge_sums(acc, 0)
        if rank == 0:
            ctx.tonemap(acc.data_ptr(), spp_job, width, height, rgb8.data_ptr(), stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # one instrumented step: exact traversal counts for the algorithmic-bytes model (not timed)
    step(L.FLAG_COUNT_TRAVERSAL)
    counted = scene.collect_stats(stream)
    fence()

    kernel_ms, launches, segments, trav_ms, trav_launches, mesh_rays = 0.0, 0, 0, 0.0, 0, 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(L.FLAG_TIME_KERNELS)             # HIP events around every closest-hit launch, on `stream`
        st = scene.collect_stats(stream)
        kernel_ms += st["ms_kernels"]; launches += st["launches"]; segments += st["segments"]
        trav_ms += st["ms_traverse"]; trav_launches += st["traverse_launches"]; mesh_rays += st["mesh_rays"]
    fence()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed, float(segments)], dtype=torch.float64, device=device)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    total_segments = float(t[1])

    if rank == 0:
        seg_per_step = counted["segments"]
        if trav_launches:       # two-kernel pipeline: the traversal kernel dominates
            kernel, unit_name = "traverse_kernel", "ray"
            units = mesh_rays
            nodes_per_unit = counted["nodes_visited"] / max(1, counted["mesh_rays"])
            tris_per_unit = counted["prims_tested"] / max(1, counted["mesh_rays"])
            bytes_per_unit = TRAVERSE_RAY_BYTES + nodes_per_unit * NODE_BYTES + tris_per_unit * TRI_BYTES
            k_ms, k_launches = trav_ms, trav_launches
        else:
            kernel, unit_name = "bounce_kernel", "segment"
            units = segments
            nodes_per_unit = counted["nodes_visited"] / max(1, seg_per_step)
            tris_per_unit = counted["prims_tested"] / max(1, seg_per_step)
            bytes_per_unit = 2 * STATE_BYTES + nodes_per_unit * NODE_BYTES + tris_per_unit * TRI_BYTES
            k_ms, k_launches = kernel_ms, launches
        avg_launch_ms = k_ms / max(1, k_launches)
        units_per_launch = units / max(1, k_launches)
        achieved_gbs = (units_per_launch * bytes_per_unit) / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        flops_per_unit = nodes_per_unit * NODE_FLOPS + tris_per_unit * TRI_FLOPS + (0 if trav_launches else SHADE_FLOPS)
        valu_tflops = (units_per_launch * flops_per_unit) / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
        traffic = None
        tfile = ROOT / "profiles" / "hbm_traffic.json"
        if tfile.exists():   # HBM bytes per launch from the committed rocprofv3 PMC passes of this workload
            try:
                traffic = json.loads(tfile.read_text()).get(args.workload, {}).get(kernel, {}).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mpath-segments/sec (whole node)",
            "value": total_segments / elapsed / 1e6,
            "unit": "Mseg/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if args.partition == "samples" else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.workload, "width": width, "height": height, "spp_per_gpu": spp if args.partition == "samples" else None,
                       "spp_job": spp_job, "max_depth": depth, "integrator": "RandomWalk", "hit_rule": args.rule,
                       "partition": args.partition, "mesh_triangles": scene.info()["n_mesh_triangles"],
                       "segments_per_step_per_gpu": seg_per_step},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel, "launches": k_launches, "avg_launch_ms": avg_launch_ms,
                         "unit_of_work": unit_name, "bytes_per_unit": bytes_per_unit, "nodes_per_unit": nodes_per_unit,
                         "triangles_per_unit": tris_per_unit, "units_per_launch": units_per_launch,
                         "all_kernels_ms_per_step": kernel_ms / args.steps, "all_launches": launches,
                         "note": "algorithmic bytes charge every BVH node and triangle fetch to memory; when the scene is "
                                 "cache-resident they are cache hits, so frac can exceed 1 while `traffic` (HBM bytes per "
                                 "launch from the committed PMC passes) stays far below it",
                         # the other term of the model: FP64-equivalent vector flops of the same launches
                         "valu": {"flops_per_unit": flops_per_unit, "achieved": valu_tflops, "peak": VALU_F64_PEAK_TFLOPS,
                                  "unit": "TFLOP/s", "frac": valu_tflops / VALU_F64_PEAK_TFLOPS}},
        }
        if not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")), 16)
            cpu_spp = args.cpu_spp or (1 if w["recipe"] == "synthetic_mesh_scene" else max(1, min(spp, 16)))
            if w["recipe"] == "synthetic_mesh_scene":
                out["cpu_baseline"] = None  # linear scans over 1M triangles: minutes per 1e5 segments; not run by default
            else:
                v, secs, segs = cpu_baseline(w, hit_rule, cpu_spp, threads)
                out["cpu_baseline"] = {"value": v, "unit": "Mseg/s", "cores": threads, "kind": "port",
                                       "sample": "same scene and camera, samples [0,%d) of %d per pixel: %d segments in %.1f s"
                                                 % (cpu_spp, spp, segs, secs)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
