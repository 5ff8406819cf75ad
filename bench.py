"""Headline benchmark: Mrays/s and samples/s of the wavefront path tracer on MI355X.

python bench.py --gpus N --steps K --warmup W
One step = one pass of the hot path over one batch: --spp-per-step frames of the workload at full resolution.
For N > 1 the driver launches one process per GPU (torch.distributed, backend "nccl" = RCCL); every rank renders
its own interleaved slice of frame indices (weak scaling: per-GPU work is fixed) and the accumulation targets are
summed with one all-reduce inside the timed region. Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DATA_ROOT = os.path.join(ROOT, "tests", "golden")
HBM_PEAK_GBS = 8000.0

# workload -> (scene name, texture directory, quicksave below data/saves, label, default triangle budget)
WORKLOADS = {
    "cornell": ("cornell_box", "cornell_box_textures", "cornell_box/default.rt_save", "Cornell box", None),
    "city": ("Bistro_outside", "Bistro_textures", "bistro/outside.rt_save", "Generated city block (Bistro exterior stand-in, the reference's Bistro lights and camera)", 2800000),
    "interior": ("Arcade", "Arcade_textures", "Arcade/default.rt_save", "Generated hall with 32 lights (Arcade stand-in)", 1000000),
}


def workload_root(args, rank, barrier=None):
    """The directory that holds data/ for the workload; generated scenes are built once per box by rank 0."""
    if args.workload == "cornell":
        return DATA_ROOT
    from path_tracer_amd.tools import procedural
    triangles = args.triangles or WORKLOADS[args.workload][4]
    root = os.path.join(args.scene_cache, "%s_%d%s" % (args.workload, triangles, "_t%d" % args.texture_size if args.texture_size else ""))
    if rank == 0:
        options = {"triangles": triangles}
        if args.texture_size:
            options["texture_size"] = args.texture_size
        if args.workload == "city":
            options["lights_file"] = os.path.join(DATA_ROOT, "data", "Bistro_outside.lights")
            options["save_file"] = os.path.join(DATA_ROOT, "data", "saves", "bistro", "outside.rt_save")
        begin = time.perf_counter()
        procedural.ensure_scene(args.workload, root, **options)
        print("workload %s ready under %s (%.1f s)" % (args.workload, root, time.perf_counter() - begin), file=sys.stderr)
    if barrier is not None:
        barrier()
    return root


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=32)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--spp-per-step", type=int, default=32)
    p.add_argument("--path-length", type=int, default=4)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--force-collectives", action="store_true", help="take the multi-GPU code path (process group, all-reduce) even with one rank")
    p.add_argument("--workload", choices=sorted(WORKLOADS), default="cornell",
                   help="cornell: the reference's own scene file; city / interior: generated stand-ins for Bistro exterior / Arcade")
    p.add_argument("--triangles", type=int, default=None, help="triangle budget of a generated workload")
    p.add_argument("--texture-size", type=int, default=None, help="edge length of the textures of a generated workload (default 256)")
    p.add_argument("--lights", type=int, default=None,
                   help="replace the workload's lights by this many spheres placed uniformly in the scene's bounding box (seed 4, radius 0.1 to 0.5)")
    p.add_argument("--light-sampling", choices=("loop", "tree"), default="loop",
                   help="loop: the reference's loops over all lights per vertex; tree: one descent through the light tree")
    p.add_argument("--scene-cache", default=os.path.join(os.environ.get("TMPDIR", "/tmp"), "pt_scenes"))
    return p.parse_args()


def trace_bytes_per_ray(kind, nodes_per_ray, tris_per_ray, node_bytes, tri_bytes):
    """Algorithmic bytes one traversal kernel moves per ray: its ray and result records plus every node and
    triangle the ray visits at full size, whether the fetch is served by LDS, a cache or HBM."""
    if kind == "extension":
        stream = 32 + 16  # ray record read, hit record written
    else:
        stream = 48 + 32 + 1  # shadow record read, radiance read-modify-write, material byte
    return stream + nodes_per_ray * node_bytes + tris_per_ray * tri_bytes


def measured_counters(workload, kernel):
    """Per-launch hardware counters of a kernel from the committed counter passes (rocprofv3 --pmc runs of this
    benchmark, scripts/gpu_pmc.sh + scripts/pmc_to_json.py), or None. Counters cannot be read inside the timed run."""
    path = os.path.join(ROOT, "profiles", "pmc_current.json")
    if not os.path.exists(path):
        return None
    with open(path) as file:
        return json.load(file).get(workload, {}).get("kernels", {}).get(kernel)


def generated_lights(count, factor, summand):
    """count lights uniform in the bounding box of the quantised positions (21 bits per axis)."""
    rng = np.random.default_rng(4)
    low = np.asarray(summand, np.float64)
    high = low + np.asarray(factor, np.float64) * float((1 << 21) - 1)
    lights = np.zeros((count, 4), np.float32)
    lights[:, :3] = rng.uniform(low, high, (count, 3))
    lights[:, 3] = rng.uniform(0.1, 0.5, count)
    return lights


def cpu_baseline(args, threads, root, first_frame_index):
    """The CPU oracle on a bounded sample of the same workload (reported baseline, not the target)."""
    import oracle
    from oracle import formats
    data = os.path.join(root, "data")
    name, textures, save, label, _ = WORKLOADS[args.workload]
    scene = formats.load_scene(os.path.join(data, name + ".vks"), os.path.join(data, textures))
    lights = formats.load_lights(os.path.join(data, name + ".lights"))
    spec = formats.load_rt_save(os.path.join(data, "saves", save))
    if args.lights is not None:
        lights = generated_lights(args.lights, scene.dequantization_factor, scene.dequantization_summand)
    # About 10-30 s of CPU work: one step of the Cornell workload, a quarter of one for the larger scenes
    width, height, spp = args.width, args.height, (32 if args.workload == "cornell" else 8)
    constants = oracle.write_constants(spec, width, height, scene, lights)
    cpu_scene = oracle.OracleScene(scene)
    begin = time.perf_counter()
    _, rays = cpu_scene.render(constants, lights, strategy=3, path_length=args.path_length, first_frame_index=first_frame_index, frame_count=spp, threads=threads,
                               light_sampling=int(args.light_sampling == "tree"))
    seconds = time.perf_counter() - begin
    return {"value": rays / seconds * 1e-6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "CPU oracle, %s %dx%d, %d spp, NEE, path length %d, %d rays in %.2f s" % (label, width, height, spp, args.path_length, rays, seconds)}


def main():
    args = parse_args()
    # The C host reports progress on stdout like the reference does; keep stdout for the one JSON line
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    from path_tracer_amd import hip_api, host_api, multi_gpu

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU fallback")
    torch.cuda.set_device(local_rank)
    distributed = world_size > 1 or args.force_collectives
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))

    root = workload_root(args, rank, dist.barrier if distributed else None)
    name, _, save, label, _ = WORKLOADS[args.workload]
    spec = host_api.quickload(os.path.join(root, "data", "saves", save))
    renderer = host_api.Renderer(spec, root, device=local_rank, width=args.width, height=args.height,
                                 sampling_strategy=hip_api.SAMPLING_STRATEGY_NEE, path_length=args.path_length)
    constants = renderer.constants()
    lib = hip_api.lib()
    import ctypes as C
    light_count = renderer.lit_scene.spherical_light_count
    if args.lights is not None:
        header = renderer.lit_scene.scene.header
        lights = generated_lights(args.lights, list(header.dequantization_factor), list(header.dequantization_summand))
        if lib.pt_set_lights(renderer.device_scene, lights.ctypes.data_as(C.c_void_p), len(lights)):
            raise RuntimeError("pt_set_lights failed")
        light_count = len(lights)
    renderer.set_light_sampling(hip_api.LIGHT_SAMPLING_TREE if args.light_sampling == "tree" else hip_api.LIGHT_SAMPLING_EXACT)

    spp = args.spp_per_step
    next_frame = [spec.frame_index + 1]

    def render_step(accumulate):
        # Sample slices: rank r renders frames base + r, base + r + G, ...
        first, count, stride = multi_gpu.frame_slice(rank, world_size, next_frame[0], spp * world_size)
        code = lib.pt_render_samples(renderer.device_scene, C.byref(constants), first, count, stride, int(accumulate))
        if code:
            raise RuntimeError("pt_render_samples failed")
        next_frame[0] += spp * world_size

    def barrier():
        renderer.synchronize()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    # One instrumented step for the per-ray node and triangle counts (deterministic given the frame indices)
    renderer.set_instrumentation(kernel_timing=False, visit_counters=True)
    renderer.reset_statistics()
    render_step(False)
    renderer.synchronize()
    profile = renderer.statistics()
    renderer.set_instrumentation(kernel_timing=False, visit_counters=False)
    for _ in range(args.warmup):
        render_step(True)
    tensor = multi_gpu.accumulation_tensor(renderer)
    if distributed:
        dist.all_reduce(tensor)  # warm up the communicator
    renderer.set_instrumentation(kernel_timing=True, visit_counters=False)
    renderer.reset_statistics()
    barrier()
    begin = time.perf_counter()
    for step in range(args.steps):
        render_step(step != 0)
    multi_gpu.all_reduce_accumulation(renderer, total_sample_count=args.steps * spp * world_size)
    barrier()
    seconds = time.perf_counter() - begin
    stats = renderer.statistics()
    rays = stats.extension_rays + stats.shadow_rays
    totals = torch.tensor([float(rays), float(stats.paths), seconds], dtype=torch.float64, device="cuda")
    if distributed:
        maximum = totals.clone()
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        dist.all_reduce(maximum, op=dist.ReduceOp.MAX)
        seconds = float(maximum[2])
    total_rays, total_paths = float(totals[0]), float(totals[1])

    if rank == 0:
        info = renderer.bvh_info()
        kinds = (("extension", 1, profile.extension_rays, 0), ("shadow", 2, profile.shadow_rays, 1))
        kernels = {}
        for name, index, profile_rays, slot in kinds:
            launches = stats.kernel_launches[index]
            timed_rays = stats.extension_rays if name == "extension" else stats.shadow_rays
            if not launches or not profile_rays:
                continue
            nodes = profile.nodes_visited[slot] / profile_rays
            tris = profile.triangles_tested[slot] / profile_rays
            bytes_per_ray = trace_bytes_per_ray(name, nodes, tris, info.node_bytes, info.triangle_bytes)
            avg_ms = stats.kernel_ms[index] / launches
            kernels[name] = {"total_ms": stats.kernel_ms[index], "launches": int(launches), "avg_ms": avg_ms, "rays_per_launch": timed_rays / launches,
                             "nodes_per_ray": nodes, "tris_per_ray": tris, "bytes_per_ray": bytes_per_ray,
                             "achieved_gbs": bytes_per_ray * timed_rays / launches / (avg_ms * 1e-3) * 1e-9}
        dominant = max(kernels, key=lambda k: kernels[k]["total_ms"])
        d = kernels[dominant]
        # Counters exist for the configurations that were profiled: the default batch of the workloads as they are
        profiled = spp == 32 and not args.triangles and args.lights is None and not args.texture_size and args.light_sampling == "loop" \
            and (args.width, args.height, args.path_length) == (1920, 1080, 4)
        counters = measured_counters(args.workload, "k_trace_%s<false>" % dominant) if profiled else None
        result = {
            "metric": "Mrays/s", "value": total_rays / seconds * 1e-6, "unit": "Mrays/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": seconds / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": ("Cornell box scene file of the reference (fixture copy of data/cornell_box.vks, its textures, lights and quicksave)" if args.workload == "cornell"
                     else "synthetic: scene generated by path_tracer_amd/tools/procedural.py in the reference's file formats"),
            "config": {"workload": "%s %dx%d, NEE + MIS, path length %d, %d spp per step per GPU, frames %d.." % (label, args.width, args.height, args.path_length, spp, spec.frame_index + 1),
                       "lights": int(light_count), "light_sampling": args.light_sampling,
                       "spp_total": args.steps * spp * world_size, "parallelism": "sample slices x%d + all-reduce" % world_size},
            "samples_per_s": total_paths / seconds,
            "rays_per_path": total_rays / total_paths,
            "roofline": {"bound": "hbm", "kernel": "k_trace_%s" % dominant, "achieved": d["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": d["achieved_gbs"] / HBM_PEAK_GBS,
                         "traffic": counters["hbm_bytes_per_launch"] if counters else None,
                         "algorithmic_bytes_per_launch": d["bytes_per_ray"] * d["rays_per_launch"],
                         "avg_launch_ms": d["avg_ms"], "rays_per_launch": d["rays_per_launch"], "bytes_per_ray": d["bytes_per_ray"],
                         "nodes_per_ray": d["nodes_per_ray"], "triangles_per_ray": d["tris_per_ray"],
                         # What the counters say about the kernel besides bytes: VALU instructions per launch (per wave) and
                         # the share of lanes active in them; a SIMD issues one wave instruction every two cycles (32 lanes
                         # a cycle), so at 2.4 GHz the 1024 SIMDs issue at most 1.23e12 a second
                         "valu_instructions_per_launch": counters["valu_instructions_per_launch"] if counters else None,
                         "valu_lane_utilisation": counters["valu_lane_utilisation"] if counters else None,
                         "valu_issue_frac": (counters["valu_instructions_per_launch"] / (d["avg_ms"] * 1e-3) / 1.2288e12) if counters else None,
                         "l2_hit_rate": counters["l2_hit_rate"] if counters else None},
            "kernel_ms": {name: stats.kernel_ms[i] for i, name in enumerate(hip_api.KERNEL_NAMES)},
            "bvh": {"nodes": info.node_count, "triangles": info.triangle_count, "depth": info.depth, "build_ms": info.build_ms},
        }
        if not args.no_cpu_baseline and world_size == 1:
            result["cpu_baseline"] = cpu_baseline(args, min(16, os.cpu_count() or 1), root, spec.frame_index + 1)
        else:
            result["cpu_baseline"] = None
        C.CDLL(None).fflush(None)
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    renderer.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
