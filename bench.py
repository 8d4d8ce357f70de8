#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X: Mpath-segments/s, whole job.

One "step" = one full render of the workload (every pixel, every sample, depth-10 random walk)
from a scene that is already resident in HBM, including the merge of the per-rank buffers and the
tone mapping. With N > 1 ranks (torch.distributed over RCCL, one process per GPU) every rank
renders its own `spp` samples per pixel of an N*spp job (sample-range partition, weak scaling) and
one reduce merges the per-pixel XYZ sums. Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python bench.py --gpus 8            # starts its own 8 ranks (or run it under torch.distributed.run)

Order of a run: the CPU baseline (rank 0 of a 1-GPU run only), W untimed warm-up steps, K timed steps with no
instrumentation at all (one fence on either side), then untimed instrumented steps (HIP events around every launch;
exact traversal counts) from which the roofline terms are formed.
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, ~8 TB/s
N_SIMD, WAVE = 256 * 4, 64   # 256 CUs x 4 SIMDs, 64 lanes per wave
MAX_CLOCK_GHZ = 2.4
L2_BYTES, INFINITY_CACHE_BYTES = 32 << 20, 256 << 20
TRI_BYTES = 80                      # a mesh triangle (f64 vertices + file-order index); a BVH node is 32 or 48 bytes (scene info)
# Algorithmic bytes per unit of the two kernels a roofline is quoted for:
#   traverse_kernel, per listed ray: list entry (4) + ray origin and direction (48) + the hit record it folds its result
#     into (12 read, 28 written) + the BVH nodes (at the scene's real node size) and triangles that ray visits
#   bounce_kernel (scenes without meshes), per segment: one path-state record read and one written (6 f64 + 4 u32 + 16
#     chain bytes each)
TRAVERSE_RAY_BYTES = 4 + 48 + 12 + 28
STATE_BYTES = 80
# flop model of the same units (FP64-equivalent): 40 per node visit, 55 per triangle test, 180 + 45 per analytic object
NODE_FLOPS, TRI_FLOPS, SHADE_FLOPS = 40, 55, 180 + 6 * 45
VALU_F64_PEAK_TFLOPS = 78.6

WORKLOADS = {
    # the headline configuration: Suzanne (968 triangles) + emitter + ground spheres, 1920x1080, 64 spp
    "suzanne_1080p_64spp": dict(recipe="scene2", width=1920, height=1080, spp=64, depth=10),
    "suzanne_specular_1080p_64spp": dict(recipe="scene2", width=1920, height=1080, spp=64, depth=10, specular_mesh=True),
    "scene1_400x225_100spp": dict(recipe="scene1", width=400, height=225, spp=100, depth=10),
    "scene2_850x480_100spp": dict(recipe="scene2", width=850, height=480, spp=100, depth=10),
    "scene3_720p_100spp": dict(recipe="scene3", width=1280, height=720, spp=100, depth=10),
    # the 1 048 576-triangle procedural mesh at 4K: the full 256 spp, and a 16 spp cut for quick runs
    "mesh1m_4k_256spp": dict(recipe="synthetic_mesh_scene", width=3840, height=2160, spp=256, depth=10),
    "mesh1m_4k_16spp": dict(recipe="synthetic_mesh_scene", width=3840, height=2160, spp=16, depth=10),
    # scene3's layout with glossy / dielectric extension materials, depth 32: the full 1024 spp and a 16 spp cut
    "scene3_extended_4k_1024spp": dict(recipe="scene3_extended", width=3840, height=2160, spp=1024, depth=32),
    "scene3_extended_4k_16spp": dict(recipe="scene3_extended", width=3840, height=2160, spp=16, depth=32),
}


def build_scene(ctx, w, hit_rule, from_rgb=None, library=True, scene_flags=0, **override):
    from raytracing_amd import api, scenes
    kw = {k: v for k, v in w.items() if k in ("width", "height", "specular_mesh")}
    kw.update(override)
    builder, cam_args = getattr(scenes, w["recipe"])(from_rgb=from_rgb, **kw)
    if library:
        return api.Scene(ctx, *builder.arrays(), hit_rule=hit_rule, flags=scene_flags), api.camera_new(*cam_args)
    return builder, cam_args


def cpu_baseline(w, hit_rule, threads, target_seconds=10.0):
    """The CPU oracle (linear scans over objects and triangles: the reference's algorithm) on a bounded sample of the
    same workload, on this machine's host cores. The sample keeps the camera and the scene and cuts samples per pixel
    first, then (for the million-triangle mesh, where one segment costs a million triangle tests) the resolution: the
    same field of view through fewer pixels is a uniform subsample of the image's paths.
    Two shorter legs beside it: the same port on one thread, and the same port with a median-split BVH over each mesh
    (not the reference's algorithm -- the "fair" CPU figure: same results, same kind of acceleration as the GPU path)."""
    import _oracle as O
    mesh = w["recipe"] == "synthetic_mesh_scene"
    div = 60 if mesh else 1
    override = dict(width=w["width"] // div, height=w["height"] // div) if div > 1 else {}
    builder, cam_args = build_scene(None, w, hit_rule, from_rgb=O.from_rgb, library=False, **override)
    cam = O.camera(*cam_args)
    where = " through %dx%d pixels (1/%d of the image's)" % (cam.width, cam.height, div * div) if div > 1 else ""

    def leg(scene, n_threads, seconds, first_spp=1):
        spp, total_seg, total_s, done = first_spp, 0, 0.0, 0
        # grow the sample until it has run for about `seconds` (each round renders samples [done, done + spp))
        while done < w["spp"] and total_s < seconds:
            spp = min(spp, w["spp"] - done)
            _, _, _, st = scene.render(cam, O.INT_RANDOM_WALK, w["depth"], w["spp"], seed=0, sample_range=(done, done + spp),
                                       threads=n_threads, want_lin=False)
            total_seg += st.segments; total_s += st.seconds; done += spp
            if st.seconds > 0:
                spp = max(1, min(int(spp * min(4.0, 0.5 * seconds / st.seconds)), 64))
        return total_seg / total_s / 1e6, "same scene and camera%s, samples [0,%d) of %d per pixel: %d segments in %.1f s" % (
            where, done, w["spp"], total_seg, total_s)

    scene = O.Scene(*builder.arrays(), hit_rule=hit_rule)
    value, what = leg(scene, threads, target_seconds)
    extra = {}
    if not mesh:   # one thread of the linear scan over a million triangles would take minutes for a handful of paths
        v1, what1 = leg(scene, 1, 0.3 * target_seconds)
        extra["one_thread"] = {"value": v1, "unit": "Mseg/s", "cores": 1, "sample": what1}
    scene.enable_bvh()
    vb, whatb = leg(scene, threads, 0.4 * target_seconds, first_spp=1 if not mesh else 4)
    extra["with_bvh"] = {"value": vb, "unit": "Mseg/s", "cores": threads, "sample": whatb,
                         "note": "not the reference's algorithm: the port with a median-split BVH per mesh, same results"}
    return value, what, extra


# ------------------------------------------------------------------------------------------- launcher

def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """`python bench.py --gpus N` typed without a launcher: start N fresh rank processes (one per GPU) and wait.
    This parent never touches the GPU or imports torch; rank 0 prints the JSON line on the stdout it inherits."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env))
    rc = 0
    live = set(range(n))
    while live:
        time.sleep(0.2)
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (r, code))
                for o in live:
                    procs[o].terminate()
    return rc


# ------------------------------------------------------------------------------------------- roofline

def committed_counters(workload, rule, kernel):
    """Per-launch hardware counters of `kernel` from the rocprofv3 PMC passes of this workload committed under
    profiles/ (tools/pmc_summary.py writes the file): HBM bytes, VALU-busy and lane-utilisation fractions."""
    f = ROOT / "profiles" / "pmc_summary.json"
    try:
        d = json.loads(f.read_text())
        return d.get("%s/%s" % (workload, rule), {}).get(kernel)
    except Exception:
        return None


def running_source_hash():
    try:
        sys.path.insert(0, str(ROOT / "tools"))
        from source_hash import source_hash
        return source_hash()
    except Exception:
        return None


def roofline(args, w, scene_info, counted, timed, world):
    """Roofline of the dominant kernel (traverse_kernel with meshes, bounce_kernel without), per launch.

    The headline figures (`achieved`, `peak`, `frac`, `bound`) are ALGORITHMIC: the per-unit model of the work a launch
    has to do, times the units the launch processed, over the launch's live duration (HIP events on the stream the
    kernel runs on). Two model terms exist and the one that can bind is quoted:
      algorithmic_flops  FP64-equivalent flops per unit (40 per node visit, 55 per triangle test, plus shading and the
                         analytic objects for the fused kernel) against the 78.6 TFLOP/s vector-FP64 peak -> bound "valu"
      algorithmic_bytes  bytes per unit with every node, triangle and record fetch charged to memory, against 8 TB/s ->
                         bound "hbm". It is a roof only when the tree exceeds the 256 MiB Infinity Cache (otherwise
                         node fetches are served from LDS or cache and the term can exceed 1); it is always printed.
    Beside them, from the committed rocprofv3 counter passes of the same workload (profiles/pmc_summary.json), divided
    by the same live launch time -- how busy the hardware was, whatever it was busy with:
      hbm_counters         2 * FETCH_SIZE + WRITE_SIZE per launch against 8 TB/s (`traffic` is this byte count)
      valu_issue_counters  executed lane-instructions per launch (SQ_THREAD_CYCLES_VALU) against one wave64 vector
                           instruction per SIMD per 4 cycles at the maximum clock = valu_busy x lane_utilisation
    `counters_from` names the file, or says "stale" when the profiled sources are not the ones being timed.
    """
    two_kernel = timed["traverse_launches"] > 0
    if two_kernel:
        kernel, unit_name = "traverse_kernel", "ray"
        units = timed["mesh_rays"]
        per_unit = max(1, counted["mesh_rays"])
        base_bytes, k_ms, k_launches = TRAVERSE_RAY_BYTES, timed["ms_traverse"], timed["traverse_launches"]
    else:
        kernel, unit_name = "bounce_kernel", "segment"
        units = timed["segments"]
        per_unit = max(1, counted["segments"])
        base_bytes, k_ms, k_launches = 2 * STATE_BYTES, timed["ms_kernels"], timed["launches"]
    node_bytes = scene_info["node_bytes"] / scene_info["n_nodes"] if scene_info.get("n_nodes") else 0.0
    nodes_per_unit = counted["nodes_visited"] / per_unit
    tris_per_unit = counted["prims_tested"] / per_unit
    bytes_per_unit = base_bytes + nodes_per_unit * node_bytes + tris_per_unit * TRI_BYTES
    avg_ms = k_ms / max(1, k_launches)
    secs = avg_ms * 1e-3
    units_per_launch = units / max(1, k_launches)
    algorithmic_gbs = units_per_launch * bytes_per_unit / secs / 1e9 if secs > 0 else 0.0
    flops_per_unit = nodes_per_unit * NODE_FLOPS + tris_per_unit * TRI_FLOPS + (0 if two_kernel else SHADE_FLOPS)
    model_tflops = units_per_launch * flops_per_unit / secs / 1e12 if secs > 0 else 0.0
    tree_bytes = scene_info["node_bytes"] + scene_info["triangle_bytes"]
    bytes_bind = tree_bytes > INFINITY_CACHE_BYTES
    terms = {"algorithmic_flops": {"achieved": model_tflops, "peak": VALU_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": model_tflops / VALU_F64_PEAK_TFLOPS, "flops_per_unit": flops_per_unit},
             "algorithmic_bytes": {"achieved": algorithmic_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algorithmic_gbs / HBM_PEAK_GBS,
                                   "bytes_per_unit": bytes_per_unit, "node_bytes": node_bytes, "counts_as_hbm_roof": bytes_bind,
                                   "tree_bytes": tree_bytes}}
    pmc = committed_counters(args.workload, args.rule, kernel)
    traffic = None
    counters_from = None
    if pmc and secs > 0:
        running = running_source_hash()
        profiled = pmc.get("source_hash")
        counters_from = "profiles/pmc_summary.json"
        if profiled is None or running is None or profiled != running:
            counters_from = "stale: profiles/pmc_summary.json was taken from sources %s, this run times %s" % (profiled, running)
        traffic = pmc.get("hbm_bytes_per_launch")
        if traffic is not None:
            gbs = traffic / secs / 1e9
            terms["hbm_counters"] = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                     "l2_hit_rate": pmc.get("l2_hit_rate")}
        if pmc.get("lane_instructions_per_launch"):
            # the peak is taken at the chip's maximum clock: the live launches are not the profiled ones, and a peak at the
            # (lower) clock of the profiled run would flatter a launch that ran faster live
            clock = MAX_CLOCK_GHZ
            peak = N_SIMD * WAVE * clock / 4.0 / 1e3                       # T lane-instructions / s
            ach = pmc["lane_instructions_per_launch"] / secs / 1e12
            terms["valu_issue_counters"] = {"achieved": ach, "peak": peak, "unit": "Tlane-instr/s", "frac": ach / peak,
                                            "valu_busy": pmc.get("valu_busy"), "lane_utilisation": pmc.get("lane_utilisation"),
                                            "lane_instructions_per_unit": pmc["lane_instructions_per_launch"] / max(1.0, units_per_launch),
                                            "peak_clock_ghz": clock, "profiled_clock_ghz": pmc.get("clock_ghz"),
                                            "profiled_launch_ms": pmc.get("launch_ms")}
    bound = "algorithmic_bytes" if bytes_bind and terms["algorithmic_bytes"]["frac"] >= terms["algorithmic_flops"]["frac"] else "algorithmic_flops"
    b = terms[bound]
    issue = terms.get("valu_issue_counters", {}).get("frac")
    hbm = terms.get("hbm_counters", {}).get("frac")
    return {"bound": "hbm" if bound == "algorithmic_bytes" else "valu", "achieved": b["achieved"], "peak": b["peak"],
            "unit": b["unit"], "frac": b["frac"], "traffic": traffic, "binding_term": bound,
            "counter_issue_frac": issue, "counter_hbm_frac": hbm,
            "kernel": kernel, "launches": k_launches, "avg_launch_ms": avg_ms, "unit_of_work": unit_name,
            "units_per_launch": units_per_launch, "nodes_per_unit": nodes_per_unit, "triangles_per_unit": tris_per_unit,
            "flops_per_unit": flops_per_unit, "bytes_per_unit": bytes_per_unit,
            "all_kernels_ms_per_step": timed["ms_kernels"] / max(1, timed["steps"]),
            "all_launches": timed["launches"], "terms": terms, "counters_from": counters_from}


# ------------------------------------------------------------------------------------------- one rank

def dry_run(args, world, rank):
    """The launcher, the rendezvous and the merge without a GPU (gloo): every rank contributes rank + 1 to its share of
    a small buffer, rank 0 checks the sum and prints a line of the usual shape."""
    import torch
    import torch.distributed as dist
    from raytracing_amd import parallel
    if world > 1:
        dist.init_process_group("gloo")
    n = 96 * 54 * 3
    t0 = time.perf_counter()
    acc = torch.full((n,), float(rank + 1), dtype=torch.float32)
    spp_job = 64 * world if args.scaling == "weak" and args.partition == "samples" else 64
    share = parallel.rank_share(args.partition, spp_job, rank, world)
    parallel.merge_sums(acc, 0)
    if world > 1:
        dist.barrier()
    ms = (time.perf_counter() - t0) * 1e3
    per_rank = [None] * world
    if world > 1:
        dist.all_gather_object(per_rank, {"rank": rank, "ms": ms, "share": share})
    else:
        per_rank = [{"rank": 0, "ms": ms, "share": share}]
    if rank == 0:
        ok = bool((acc == world * (world + 1) / 2).all())
        print(json.dumps({"metric": "Mpath-segments/sec (whole node)", "value": 0.0, "unit": "Mseg/s", "n_gpus": world, "dry_run": True,
                          "world_size": dist.get_world_size() if world > 1 else 1, "merge_ok": ok, "per_rank": per_rank,
                          "scaling": "weak" if args.partition == "samples" and args.scaling == "weak" else "strong", "spp_job": spp_job,
                          "steps": args.steps, "warmup": args.warmup}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def in_process_devices(args, w):
    """One process, several GPUs behind the C ABI: rtmi_multi_render partitions the render over the devices (sample ranges or
    interleaved tiles), runs one host thread per device and merges the sums with one ncclReduce inside the library -- the
    path the Rust shim calls. Strong scaling (the workload's spp are split). The call returns the image on the host, so
    unlike the rank path the time includes the copy of the 8-bit image back (W * H * 3 bytes)."""
    import time as _t
    from raytracing_amd import api, scenes, _lib as L
    n = args.devices
    ids = [0] * n if args.share_gpu else list(range(n))
    hit_rule = L.HIT_FIRST_INDEX if args.rule == "first_index" else L.HIT_NEAREST
    kw = {k: v for k, v in w.items() if k in ("width", "height", "specular_mesh")}
    builder, cam_args = getattr(scenes, w["recipe"])(from_rgb=None, **kw)
    t_build = _t.perf_counter()
    multi = api.MultiScene(ids, *builder.arrays(), hit_rule=hit_rule, max_batch_paths=args.batch_paths)
    build_s = _t.perf_counter() - t_build
    cam = api.camera_new(*cam_args)
    p = api.render_params(L.INT_RANDOM_WALK, w["depth"], w["spp"], seed=0)
    part = L.PARTITION_SAMPLES if args.partition == "samples" else L.PARTITION_TILES
    for _ in range(args.warmup):
        multi.render(cam, p, part, want_xyz=False)
    t0 = _t.perf_counter()
    segments = 0
    for _ in range(args.steps):
        _, _, st, per = multi.render(cam, p, part, want_xyz=False)
        segments += st["segments"]
    elapsed = _t.perf_counter() - t0
    info = multi.info()
    print(json.dumps({
        "metric": "Mpath-segments/sec (whole node)", "value": segments / elapsed / 1e6, "unit": "Mseg/s", "n_gpus": n,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": args.workload, "width": cam.width, "height": cam.height, "spp_job": w["spp"], "max_depth": w["depth"],
                   "integrator": "RandomWalk", "hit_rule": args.rule, "partition": args.partition, "devices": ids,
                   "scene_build_s": build_s, "image_copied_to_host": True},
        "world_size": 1, "backend": "rtmi_multi (ncclReduce inside the library)" if info["uses_rccl"] else "rtmi_multi (copies and adds)",
        "per_rank_segments": [s["segments"] for s in per], "roofline": None, "cpu_baseline": None}), flush=True)
    multi.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="suzanne_1080p_64spp", choices=sorted(WORKLOADS))
    ap.add_argument("--rule", default="first_index", choices=["first_index", "nearest"])
    ap.add_argument("--partition", default="samples", choices=["samples", "tiles"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="samples partition only: weak = every rank renders the workload's spp of an N x spp job (the default, what "
                         "the line's `scaling` says); strong = the workload's spp are split over the ranks (the 256-spp-over-8-GPUs "
                         "form of the 1 M-triangle configuration). The tile partition is always strong.")
    ap.add_argument("--devices", type=int, default=0,
                    help="one process, this many GPUs behind the C ABI (rtmi_multi_render: partition + ncclReduce inside the library) "
                         "instead of one process per GPU; prints the same line with backend \"rtmi_multi\"")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain", type=float, default=8.0,
                    help="after the timed steps, keep stepping for about this many seconds and report the rate over that stretch too "
                         "(`sustained`: clocks settled, and long enough for an outside utilisation sampler to see it); 0 = skip")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = every core this process may use)")
    ap.add_argument("--batch-paths", type=int, default=0, help="path slots per batch of launches (0 = library default)")
    ap.add_argument("--gpu-bvh", action="store_true", help="build the meshes' trees on the GPU (Morton order + radix tree) instead of the host SAH build")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: exercise the launcher, the rendezvous and the merge over gloo")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank renders on device 0 and the merge goes over gloo (RCCL refuses two "
                         "ranks on one device); the line it prints is not a scaling measurement")
    args = ap.parse_args()

    n = max(1, args.gpus)
    if n > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(n, sys.argv[1:]))       # nothing above this line has touched the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, world, rank)

    w = WORKLOADS[args.workload]
    if args.devices > 0:
        return in_process_devices(args, w)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # first, while the GPU is idle: it is the long leg of the run
        from raytracing_amd import _lib as L0
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        share = int(os.environ.get("OMP_NUM_THREADS", "0")) or cores     # the box's CPU share when it is smaller than the machine
        threads = args.cpu_threads or max(1, min(cores, share))
        rule0 = L0.HIT_FIRST_INDEX if args.rule == "first_index" else L0.HIT_NEAREST
        v, what, extra = cpu_baseline(w, rule0, threads)
        cpu = {"value": v, "unit": "Mseg/s", "cores": threads, "kind": "port", "sample": what,
               "host_cores_visible": os.cpu_count(), "cores_usable": max(1, min(cores, share))}
        cpu.update(extra)

    import torch
    import torch.distributed as dist
    from raytracing_amd import api, parallel, _lib as L

    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    stats_device = torch.device("cpu") if args.share_gpu else device     # gloo gathers host tensors
    if world > 1:
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    hit_rule = L.HIT_FIRST_INDEX if args.rule == "first_index" else L.HIT_NEAREST
    ctx = api.Context(local_rank, args.batch_paths)
    t_build = time.perf_counter()
    scene, cam = build_scene(ctx, w, hit_rule, scene_flags=L.SCENE_GPU_BVH if args.gpu_bvh else 0)
    build_s = time.perf_counter() - t_build
    width, height, spp, depth = cam.width, cam.height, w["spp"], w["depth"]
    weak = args.partition == "samples" and args.scaling == "weak"
    spp_job = spp * world if weak else spp     # weak scaling in samples: every rank its own spp of an N x spp job
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
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()                                # no instrumentation, no host synchronisation inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    last = scene.collect_stats(stream)                     # exact, from the fill counters of the last step
    seg_step, mesh_rays_step = last["segments"], last["mesh_rays"]
    segments = seg_step * args.steps

    # untimed: exact traversal counts, then HIP events around every launch (on `stream`) for the roofline terms
    step(L.FLAG_COUNT_TRAVERSAL)
    counted = scene.collect_stats(stream)
    fence()
    timed = dict(ms_kernels=0.0, launches=0, segments=0, ms_traverse=0.0, traverse_launches=0, mesh_rays=0, steps=0)
    t1 = time.perf_counter()
    for _ in range(max(1, min(args.steps, 5))):
        step(L.FLAG_TIME_KERNELS)
        st = scene.collect_stats(stream)
        for k in ("ms_kernels", "launches", "segments", "ms_traverse", "traverse_launches", "mesh_rays"):
            timed[k] += st[k]
        timed["steps"] += 1
    fence()
    instrumented_ms = (time.perf_counter() - t1) / timed["steps"] * 1e3

    t = torch.tensor([elapsed, float(segments), float(mesh_rays_step * args.steps)], dtype=torch.float64, device=stats_device)
    per_rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        gathered = [torch.zeros(1, dtype=torch.float64, device=stats_device) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([elapsed / args.steps * 1e3], dtype=torch.float64, device=stats_device))
        per_rank_ms = [float(g[0]) for g in gathered]
        elapsed = float(tmax[0])
    total_segments, total_mesh_rays = float(t[1]), float(t[2])

    # untimed as far as `value` goes: the same step for several seconds on end. `elapsed` is already the maximum over the
    # ranks, so every rank arrives at the same number of steps (the merge inside a step is a collective).
    sustained = None
    if args.sustain > 0:
        n_sus = max(1, min(20000, int(args.sustain * args.steps / max(elapsed, 1e-9))))
        fence()
        t2 = time.perf_counter()
        for _ in range(n_sus):
            step()
        fence()
        sus = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device=stats_device)
        if world > 1:
            dist.all_reduce(sus, op=dist.ReduceOp.MAX)
        sustained = {"seconds": float(sus[0]), "steps": n_sus, "ms_per_step": float(sus[0]) / n_sus * 1e3,
                     "value": total_segments / args.steps * n_sus / float(sus[0]) / 1e6, "unit": "Mseg/s"}

    if rank == 0:
        info = scene.info()
        out = {
            "metric": "Mpath-segments/sec (whole node)",
            "value": total_segments / elapsed / 1e6,
            "unit": "Mseg/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            # the segments that had a mesh to search (their ray meets a mesh's box): the traversal kernel's share of `value`
            "mesh_rays_per_step": total_mesh_rays / args.steps,
            "mesh_rays_mps": total_mesh_rays / elapsed / 1e6,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.workload, "width": width, "height": height, "spp_per_gpu": spp if weak else (spp / world if args.partition == "samples" else None),
                       "spp_job": spp_job, "max_depth": depth, "integrator": "RandomWalk", "hit_rule": args.rule,
                       "partition": args.partition, "mesh_triangles": info["n_mesh_triangles"],
                       "segments_per_step_per_gpu": seg_step, "scene_build_s": build_s,
                       "bvh_build_s": info.get("bvh_build_ms", 0.0) / 1e3,
                       "bvh_builder": "device_lbvh" if args.gpu_bvh else "host_sah"},
            "world_size": dist.get_world_size() if world > 1 else 1,
            "backend": dist.get_backend() if world > 1 else None,
            "per_rank_ms_per_step": per_rank_ms,
            "ms_per_step_instrumented": instrumented_ms,
            "sustained": sustained,
            "roofline": roofline(args, w, info, counted, timed, world),
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
