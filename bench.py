"""Headline benchmark: Mrays/s and samples/s of the wavefront path tracer on MI355X.

python bench.py --gpus N --steps K --warmup W
One step = one pass of the hot path over one batch: --spp-per-step frames of the workload at full resolution.
Two ways to use N > 1 GPUs, both with the scene replicated, the frame indices shared out in interleaved slices and
one sum of the accumulation targets over xGMI inside the timed region (its duration is also reported on its own):
* launched under torch.distributed.run (WORLD_SIZE set): one process per GPU, all-reduce of the "nccl" (= RCCL) backend;
* launched plainly as `python bench.py --gpus N`: one process drives N GPUs through the library's own multi-GPU
  entry points (pt_create_context, pt_context_render_samples, pt_context_reduce), as a C host would. With fewer than
  N GPUs visible it says so and exits with a nonzero code; it never measures fewer GPUs than it was asked for.
Two legs are timed, and both are in the line: weak scaling (--spp-per-step frames per GPU and step) and strong
scaling (--strong-frames frames per step shared by all GPUs: the same total whatever N is, at least 32 frames per GPU
up to eight GPUs). --scaling names the leg that `value` reports (weak by default). Rank 0 prints one JSON line.

The default workload is the configuration the headline metric is quoted on: the Bistro exterior at 1920x1080 with
the reference's 28 Bistro lights and its bistro/outside.rt_save camera. The Bistro mesh and textures are a separate
download of the reference and are not available, so a generated city block of 2.8 M triangles with 100 textured
materials in the reference's file formats stands in for them (--workload cornell runs the reference's own scene).
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
# A SIMD issues one VALU wave instruction every two cycles (16 lanes wide, 64-lane waves take four cycles, two
# waves alternate): 1024 SIMDs at 2.4 GHz issue at most this many wave instructions per second
VALU_PEAK_WAVE_INSTRUCTIONS_PER_S = 1024 * 2.4e9 / 2.0
MEASURED_CYCLES_PER_WAVE_INSTRUCTION = 2.4
PMC_FILE = os.path.join("profiles", "pmc_current.json")

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
    # The generator's revision is part of the name, so that a scene left by an earlier revision is not taken for this one's
    root = os.path.join(args.scene_cache, "%s_%d%s_g%d" % (args.workload, triangles, "_t%d" % args.texture_size if args.texture_size else "", procedural.REVISION))
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


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=32)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--spp-per-step", type=int, default=32)
    p.add_argument("--path-length", type=int, default=4)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--force-collectives", action="store_true", help="take the one-process-per-GPU code path (process group, all-reduce) even with one rank")
    p.add_argument("--force-context", action="store_true", help="take the one-process-many-GPUs code path (pt_create_context ... pt_context_reduce) even with one GPU")
    p.add_argument("--devices", default=None,
                   help="comma-separated GPU indices for the one-process-many-GPUs code path instead of 0..N-1 (their number must be --gpus)")
    p.add_argument("--strong-frames", type=int, default=256, help="frames per step of the strong-scaling leg, shared by all GPUs (0 skips the leg)")
    p.add_argument("--strong-steps", type=int, default=None, help="steps of the strong-scaling leg (default: a quarter of --steps, at least 2)")
    p.add_argument("--workload", choices=sorted(WORKLOADS), default="city",
                   help="city: generated stand-in for the Bistro exterior (the metric's configuration); cornell: the reference's own scene file; interior: generated stand-in for Arcade")
    p.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                   help="the leg that `value` reports. weak: --spp-per-step frames per GPU and step; strong: --strong-frames frames per step shared by all GPUs")
    p.add_argument("--cpu-threads", type=int, default=None, help="threads of the CPU baseline (default: the CPUs this process may use)")
    p.add_argument("--triangles", type=int, default=None, help="triangle budget of a generated workload")
    p.add_argument("--texture-size", type=int, default=None, help="edge length of the textures of a generated workload (default 256)")
    p.add_argument("--lights", type=int, default=None,
                   help="replace the workload's lights by this many spheres placed uniformly in the scene's bounding box (seed 4, radius 0.1 to 0.5)")
    p.add_argument("--light-sampling", choices=("loop", "tree"), default="loop",
                   help="loop: the reference's loops over all lights per vertex; tree: one descent through the light tree")
    p.add_argument("--shading-math", choices=("exact", "fast"), default="exact",
                   help="exact: correctly rounded divisions and square roots in the loops over all lights (images equal the CPU oracle bit for bit); fast: the GPU's approximate reciprocal and square root there")
    p.add_argument("--scene-cache", default=os.path.join(os.environ.get("TMPDIR", "/tmp"), "pt_scenes"))
    return p.parse_args(argv)


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
    benchmark, scripts/gpu_pmc.sh + scripts/pmc_to_json.py) and the file they were taken from, or (None, None).
    Counters cannot be collected inside the timed run: they come from separate passes over the same command."""
    path = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(path):
        return None, None
    with open(path) as file:
        entry = json.load(file).get(workload, {})
    # The traversal kernels have a second template argument (with or without skipping of what cannot matter to the
    # ray: shadow rays as decided per scene after the first batch, extension rays to the last vertex of a path). The
    # launches of a step may be of both variants, and the live timing averages over all of them: so do the counters
    # here, weighted by the launches of each variant.
    variants = [c for name, c in entry.get("kernels", {}).items() if name == kernel or name.startswith(kernel[:-1] + ",")]
    if not variants:
        return None, entry.get("source")
    launches = sum(c.get("launches", 1.0) for c in variants)
    merged = {"launches": launches}
    for key in variants[0]:
        if key != "launches" and all(key in c for c in variants):
            merged[key] = sum(c[key] * c.get("launches", 1.0) for c in variants) / launches
    return merged, entry.get("source")


def host_threads(requested=None):
    """Threads for the CPU baseline: every CPU this process may run on. On a box that hands each job a share of a
    larger machine the share is what OMP_NUM_THREADS says, not what os.cpu_count() reports for the whole machine."""
    if requested:
        return requested
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and 0 < int(share) < threads:
        threads = int(share)
    return threads


def generated_lights(count, factor, summand):
    """count lights uniform in the bounding box of the quantised positions (21 bits per axis), radii in [0.1, 0.5]:
    stream 4 of the generators' random numbers (procedural.Pcg2d)."""
    from path_tracer_amd.tools import procedural
    rng = procedural.Pcg2d(4)
    low = np.asarray(summand, np.float64)
    high = low + np.asarray(factor, np.float64) * float((1 << 21) - 1)
    lights = np.zeros((count, 4), np.float32)
    lights[:, :3] = rng.uniform(low, high, (count, 3))
    lights[:, 3] = rng.uniform(0.1, 0.5, count)
    return lights


def cpu_baseline(args, threads, root, first_frame_index):
    """The CPU oracle on a bounded sample of the same workload, best of three (reported baseline, not the target)."""
    import oracle
    from oracle import formats
    data = os.path.join(root, "data")
    name, textures, save, label, _ = WORKLOADS[args.workload]
    scene = formats.load_scene(os.path.join(data, name + ".vks"), os.path.join(data, textures))
    lights = formats.load_lights(os.path.join(data, name + ".lights"))
    spec = formats.load_rt_save(os.path.join(data, "saves", save))
    if args.lights is not None:
        lights = generated_lights(args.lights, scene.dequantization_factor, scene.dequantization_summand)
    # Three runs of 3-8 s each: eight frames of the Cornell workload, two of the larger scenes
    width, height, spp = args.width, args.height, (8 if args.workload == "cornell" else 2)
    constants = oracle.write_constants(spec, width, height, scene, lights)
    cpu_scene = oracle.OracleScene(scene)
    best, rays = None, 0
    for _ in range(3):
        begin = time.perf_counter()
        _, rays = cpu_scene.render(constants, lights, strategy=3, path_length=args.path_length, first_frame_index=first_frame_index, frame_count=spp, threads=threads,
                                   light_sampling=int(args.light_sampling == "tree"))
        seconds = time.perf_counter() - begin
        best = seconds if best is None else min(best, seconds)
    return {"value": rays / best * 1e-6, "unit": "Mrays/s", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "CPU oracle (scalar C restatement of the shader, one thread per core in use), %s %dx%d, %d spp, NEE, path length %d, %d rays, best of 3: %.2f s"
                      % (label, width, height, spp, args.path_length, rays, best)}


class Ranks:
    """One process per GPU: this rank's scene, slices by rank, all-reduce of torch.distributed (RCCL). A single
    process without a process group is the one-GPU case of the same thing."""

    def __init__(self, args, spec, root, world_size, rank, local_rank, torch=None):
        from path_tracer_amd import hip_api, host_api
        self.world_size, self.rank = world_size, rank
        # torch is plumbing for the process group only; without one it is not even imported
        self.torch = torch if (torch is not None and torch.distributed.is_initialized()) else None
        self.distributed = self.torch is not None
        self.renderer = host_api.Renderer(spec, root, device=local_rank, width=args.width, height=args.height,
                                          sampling_strategy=hip_api.SAMPLING_STRATEGY_NEE, path_length=args.path_length)
        self.renderers = [self.renderer]
        self.gpu_count = world_size
        self.parallelism = "one process per GPU, sample slices x%d + all-reduce" % world_size if self.distributed else "one GPU"

    def render(self, constants, first_frame_index, frame_count, accumulate):
        from path_tracer_amd import hip_api, multi_gpu
        import ctypes as C
        first, count, stride = multi_gpu.frame_slice(self.rank, self.world_size, first_frame_index, frame_count)
        if count and hip_api.lib().pt_render_samples(self.renderer.device_scene, C.byref(constants), first, count, stride, int(accumulate)):
            raise RuntimeError("pt_render_samples failed")

    def barrier(self):
        self.renderer.synchronize()
        if self.distributed:
            self.torch.cuda.synchronize()
            self.torch.distributed.barrier()
            self.torch.cuda.synchronize()

    def warm_up_reduction(self):
        from path_tracer_amd import multi_gpu
        if self.distributed:
            self.torch.distributed.all_reduce(multi_gpu.accumulation_tensor(self.renderer))

    def reduce(self, total_sample_count):
        """Sums the targets of all GPUs; returns the seconds it took."""
        from path_tracer_amd import multi_gpu
        self.renderer.synchronize()
        begin = time.perf_counter()
        if self.distributed:
            multi_gpu.all_reduce_accumulation(self.renderer, total_sample_count=total_sample_count)
        return time.perf_counter() - begin

    def totals(self, rays, paths, seconds, reduce_seconds):
        """(rays, paths) summed and (seconds, reduce_seconds) maximised over the ranks."""
        if not self.distributed:
            return rays, paths, seconds, reduce_seconds
        dist, torch = self.torch.distributed, self.torch
        sums = torch.tensor([float(rays), float(paths)], dtype=torch.float64, device="cuda")
        maxima = torch.tensor([seconds, reduce_seconds], dtype=torch.float64, device="cuda")
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(maxima, op=dist.ReduceOp.MAX)
        return float(sums[0]), float(sums[1]), float(maxima[0]), float(maxima[1])

    def per_gpu(self, stats_list):
        return None

    def close(self):
        self.renderer.close()
        if self.distributed:
            self.torch.distributed.destroy_process_group()


class Context:
    """One process, several GPUs through the C ABI: pt_create_context, pt_context_render_samples, pt_context_reduce."""

    def __init__(self, args, spec, root, devices):
        from path_tracer_amd import hip_api, host_api
        self.lib = hip_api.lib()
        self.renderer = host_api.Renderer(spec, root, width=args.width, height=args.height, devices=list(devices),
                                          sampling_strategy=hip_api.SAMPLING_STRATEGY_NEE, path_length=args.path_length)
        self.context = self.renderer.device_context
        self.gpu_count = int(self.lib.pt_context_device_count(self.context))
        self.devices = list(devices)
        # A view of every replica for statistics and instrumentation; replica 0 is what the renderer itself shows
        self.renderers = [self.renderer] + [host_api.ReplicaView(self.renderer, self.lib.pt_context_scene(self.context, i)) for i in range(1, self.gpu_count)]
        self.rank, self.world_size, self.distributed = 0, 1, False
        self.parallelism = "one process, %d GPU%s through pt_create_context: sample slices + pt_context_reduce (RCCL reduce over xGMI)" % (self.gpu_count, "" if self.gpu_count == 1 else "s")

    def render(self, constants, first_frame_index, frame_count, accumulate):
        import ctypes as C
        if self.lib.pt_context_render_samples(self.context, C.byref(constants), first_frame_index, frame_count, int(accumulate)):
            raise RuntimeError("pt_context_render_samples failed")

    def barrier(self):
        self.renderer.synchronize()

    def warm_up_reduction(self):
        self.renderer.reduce()

    def reduce(self, total_sample_count):
        self.renderer.reduce()
        return self.lib.pt_context_last_reduce_ms(self.context) * 1e-3

    def totals(self, rays, paths, seconds, reduce_seconds):
        return rays, paths, seconds, reduce_seconds

    def per_gpu(self, stats_list):
        return [{"device": d, "extension_rays": int(s.extension_rays), "shadow_rays": int(s.shadow_rays), "paths": int(s.paths)} for d, s in zip(self.devices, stats_list)]

    def close(self):
        self.renderer.close()


def timed_leg(backend, constants, next_frame, frames_per_step, steps, warmup, kernel_timing):
    """warmup untimed steps, then exactly `steps` steps and the sum of the targets between two barriers.
    Returns rays, paths (all GPUs), seconds and reduction seconds (maximum over ranks), and every replica's statistics."""
    def step(accumulate):
        backend.render(constants, next_frame[0], frames_per_step, accumulate)
        next_frame[0] += frames_per_step

    for _ in range(warmup):
        step(True)
    backend.warm_up_reduction()
    for r in backend.renderers:
        r.set_instrumentation(kernel_timing=kernel_timing, visit_counters=False)
        r.reset_statistics()
    backend.barrier()
    begin = time.perf_counter()
    for i in range(steps):
        step(i != 0)
    reduce_seconds = backend.reduce(steps * frames_per_step)
    backend.barrier()
    seconds = time.perf_counter() - begin
    stats_list = [r.statistics() for r in backend.renderers]
    rays = sum(s.extension_rays + s.shadow_rays for s in stats_list)
    paths = sum(s.paths for s in stats_list)
    rays, paths, seconds, reduce_seconds = backend.totals(rays, paths, seconds, reduce_seconds)
    return {"rays": rays, "paths": paths, "seconds": seconds, "reduce_seconds": reduce_seconds, "stats": stats_list,
            "frames_per_step": frames_per_step, "steps": steps}


def leg_summary(leg, gpu_count):
    return {"value": leg["rays"] / leg["seconds"] * 1e-6, "unit": "Mrays/s", "samples_per_s": leg["paths"] / leg["seconds"], "steps": leg["steps"],
            "first_frame_index": leg["first_frame_index"], "frames_per_step": leg["frames_per_step"], "frames_per_gpu_and_step": leg["frames_per_step"] / gpu_count,
            "ms_per_step": leg["seconds"] / leg["steps"] * 1e3, "all_reduce_ms": leg["reduce_seconds"] * 1e3}


def run(args, image_out=None):
    """The benchmark proper; returns the result (rank 0) or None. image_out: a list that receives the summed
    accumulation target of the leg that `value` reports and its sample count (for tests)."""
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    under_launcher = ("WORLD_SIZE" in os.environ and "RANK" in os.environ) or args.force_collectives
    if under_launcher:
        # A process that uses PyTorch and the HIP library together imports torch first, so that both share one HIP
        # (and RCCL) runtime; loaded the other way round torch finds no GPU
        import torch  # noqa: F401
    from path_tracer_amd import hip_api, host_api
    import ctypes as C
    lib = hip_api.lib()
    use_context = not under_launcher and (args.gpus > 1 or args.force_context)
    visible = int(lib.pt_get_device_count())
    if visible == 0:
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU fallback")
    if under_launcher and world_size != args.gpus and not args.force_collectives:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d: the two must agree" % (args.gpus, world_size))
    devices = list(range(args.gpus))
    if args.devices is not None:
        devices = [int(d) for d in args.devices.split(",")]
        use_context = not under_launcher
        if len(devices) != args.gpus:
            raise SystemExit("bench.py --gpus %d --devices %s: the list names %d GPUs" % (args.gpus, args.devices, len(devices)))
    if use_context and (visible < len(set(devices)) or max(devices) >= visible):
        raise SystemExit("bench.py --gpus %d: only %d GPU%s visible to this process; refusing to measure fewer GPUs than asked for" % (args.gpus, visible, " is" if visible == 1 else "s are"))
    torch = None
    if under_launcher:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world_size > 1 or args.force_collectives:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))
    root = workload_root(args, rank, torch.distributed.barrier if (torch is not None and torch.distributed.is_initialized()) else None)
    name, _, save, label, _ = WORKLOADS[args.workload]
    spec = host_api.quickload(os.path.join(root, "data", "saves", save))

    # The CPU legs come first, so that the GPU phase below is one contiguous stretch at the end of the run
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world_size == 1 and not use_context:
        cpu = cpu_baseline(args, host_threads(args.cpu_threads), root, spec.frame_index + 1)

    backend = Context(args, spec, root, devices) if use_context else Ranks(args, spec, root, world_size, rank, local_rank, torch)
    gpu_count = backend.gpu_count
    renderer = backend.renderer
    constants = renderer.constants()
    light_count = renderer.lit_scene.spherical_light_count
    if args.lights is not None:
        header = renderer.lit_scene.scene.header
        lights = generated_lights(args.lights, list(header.dequantization_factor), list(header.dequantization_summand))
        set_lights = (lambda: lib.pt_context_set_lights(backend.context, lights.ctypes.data_as(C.c_void_p), len(lights))) if use_context \
            else (lambda: lib.pt_set_lights(renderer.device_scene, lights.ctypes.data_as(C.c_void_p), len(lights)))
        if set_lights():
            raise RuntimeError("pt_set_lights failed")
        light_count = len(lights)
    renderer.set_light_sampling(hip_api.LIGHT_SAMPLING_TREE if args.light_sampling == "tree" else hip_api.LIGHT_SAMPLING_EXACT)
    shading_math = hip_api.SHADING_MATH_FAST if args.shading_math == "fast" else hip_api.SHADING_MATH_EXACT
    if (lib.pt_context_set_shading_math(backend.context, shading_math) if use_context else lib.pt_set_shading_math(renderer.device_scene, shading_math)):
        raise RuntimeError("pt_set_shading_math failed")

    spp = args.spp_per_step
    next_frame = [spec.frame_index + 1]
    # One instrumented step for the per-ray node and triangle counts (deterministic given the frame indices)
    renderer.set_instrumentation(kernel_timing=False, visit_counters=True)
    renderer.reset_statistics()
    backend.render(constants, next_frame[0], spp * gpu_count, False)
    next_frame[0] += spp * gpu_count
    backend.barrier()
    profile = renderer.statistics()
    renderer.set_instrumentation(kernel_timing=False, visit_counters=False)

    plans = [("weak", spp * gpu_count, args.steps, args.warmup)]
    if args.strong_frames > 0:
        strong_steps = args.strong_steps or (args.steps if args.scaling == "strong" else max(2, args.steps // 4))
        plans.append(("strong", args.strong_frames, strong_steps, min(args.warmup, 1)))
    elif args.scaling == "strong":
        raise SystemExit("--scaling strong needs --strong-frames > 0")
    legs = {}
    for leg_name, frames_per_step, steps, warmup in plans:
        first_timed_frame = next_frame[0] + warmup * frames_per_step
        legs[leg_name] = timed_leg(backend, constants, next_frame, frames_per_step, steps, warmup, kernel_timing=leg_name == args.scaling)
        legs[leg_name]["first_frame_index"] = first_timed_frame
        if leg_name == args.scaling and image_out is not None:
            image_out.extend(renderer.read_accumulation())
    main_leg = legs[args.scaling]
    stats = main_leg["stats"][0]
    seconds = main_leg["seconds"]
    total_rays, total_paths = main_leg["rays"], main_leg["paths"]

    result = None
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
            and (args.width, args.height, args.path_length) == (1920, 1080, 4) and args.scaling == "weak" and args.shading_math == "exact"
        counters, counter_source = measured_counters(args.workload, "k_trace_%s<false>" % dominant) if profiled else (None, None)
        # What bounds the kernel, from what the hardware counted rather than from the byte model: the HBM traffic
        # of a launch against the 8 TB/s of the part, and the VALU wave instructions of a launch against what the
        # 1024 SIMDs can issue. The larger fraction names the bound. The byte model (every node and
        # triangle visit at full size) stays beside it as a diagnostic: most of those bytes come out of LDS and L2,
        # so as a rate it exceeds what HBM could deliver.
        launch_s = d["avg_ms"] * 1e-3
        hbm_frac = valu_issue_frac = None
        if counters:
            hbm_frac = counters["hbm_bytes_per_launch"] / launch_s / (HBM_PEAK_GBS * 1e9)
            if counters.get("valu_instructions_per_launch"):
                valu_issue_frac = counters["valu_instructions_per_launch"] / launch_s / VALU_PEAK_WAVE_INSTRUCTIONS_PER_S
        if valu_issue_frac is not None and valu_issue_frac >= (hbm_frac or 0.0):
            bound = {"bound": "valu", "achieved": counters["valu_instructions_per_launch"] / launch_s * 1e-9, "peak": VALU_PEAK_WAVE_INSTRUCTIONS_PER_S * 1e-9,
                     "unit": "G wave instructions/s", "frac": valu_issue_frac}
        elif hbm_frac is not None:
            bound = {"bound": "hbm", "achieved": counters["hbm_bytes_per_launch"] / launch_s * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac}
        else:
            # No counter pass exists for this configuration: nothing measured to put a fraction on
            bound = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None}
        roofline = dict(bound)
        roofline.update({
            "kernel": "k_trace_%s" % dominant,
            "traffic": counters["hbm_bytes_per_launch"] if counters else None,
            "traffic_source": (counter_source and "%s via %s (separate rocprofv3 --pmc passes of this command, not this run)" % (counter_source, PMC_FILE)),
            "hbm_frac": hbm_frac, "valu_issue_frac": valu_issue_frac,
            # The same against what streams of independent instructions were measured to issue on this part: one wave64
            # instruction per 2.4 instead of 2 cycles of the nominal clock (scripts/valu_rates.hip,
            # profiles/r3/valu_instruction_costs.txt)
            "valu_issue_frac_of_measured_rate": valu_issue_frac * MEASURED_CYCLES_PER_WAVE_INSTRUCTION / 2.0 if valu_issue_frac is not None else None,
            "valu_lane_utilisation": counters.get("valu_lane_utilisation") if counters else None,
            "valu_instructions_per_launch": counters.get("valu_instructions_per_launch") if counters else None,
            "l2_hit_rate": counters.get("l2_hit_rate") if counters else None,
            # The vector memory path of the same counter passes: share of the kernel's cycles in which a CU's address
            # unit has work / is stalled by the vector L1, and what an L1 read request to L2 takes
            "address_unit_busy_frac": counters.get("address_unit_busy_frac") if counters else None,
            "address_unit_stalled_by_l1_frac": counters.get("address_unit_stalled_by_l1_frac") if counters else None,
            "l2_read_latency_cycles": counters.get("l2_read_latency_cycles") if counters else None,
            "l1_read_requests_per_cycle_and_cu": counters.get("l1_read_requests_per_cycle_and_cu") if counters else None,
            # The same kernel's average duration in the kernel trace that was taken on the box of the counter passes:
            # set against avg_launch_ms (this run, this box) it shows how far the two runs are apart
            "counter_run_ms": counters.get("kernel_trace_avg_ms") if counters else None,
            "avg_launch_ms": d["avg_ms"], "rays_per_launch": d["rays_per_launch"],
            "algorithmic_bytes_per_launch": d["bytes_per_ray"] * d["rays_per_launch"], "algorithmic_gbs": d["achieved_gbs"],
            "bytes_per_ray": d["bytes_per_ray"], "nodes_per_ray": d["nodes_per_ray"], "triangles_per_ray": d["tris_per_ray"]})
        summaries = {name: leg_summary(leg, gpu_count) for name, leg in legs.items()}
        result = {
            "metric": "Mrays/s", "value": total_rays / seconds * 1e-6, "unit": "Mrays/s",
            "n_gpus": gpu_count, "steps": main_leg["steps"], "warmup": args.warmup, "ms_per_step": seconds / main_leg["steps"] * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
            "data": ("Cornell box scene file of the reference (fixture copy of data/cornell_box.vks, its textures, lights and quicksave)" if args.workload == "cornell"
                     else "synthetic: scene generated by path_tracer_amd/tools/procedural.py in the reference's file formats (the reference's Bistro assets are not available)"),
            "config": {"workload": "%s %dx%d, NEE + MIS, path length %d, %d frames per step (%s), frames %d.." % (
                           label, args.width, args.height, args.path_length, main_leg["frames_per_step"],
                           "%d per GPU" % spp if args.scaling == "weak" else "shared by all GPUs", spec.frame_index + 1),
                       "lights": int(light_count), "light_sampling": args.light_sampling, "shading_math": args.shading_math,
                       "spp_total": main_leg["steps"] * main_leg["frames_per_step"], "parallelism": backend.parallelism},
            "samples_per_s": total_paths / seconds,
            "rays_per_path": total_rays / total_paths,
            "all_reduce_ms": main_leg["reduce_seconds"] * 1e3,
            "weak": summaries.get("weak"), "strong": summaries.get("strong"),
            "per_gpu": backend.per_gpu(main_leg["stats"]),
            "roofline": roofline,
            "kernel_ms": {name: stats.kernel_ms[i] for i, name in enumerate(hip_api.KERNEL_NAMES)},
            "traversal": {name: {"nodes_per_ray": k["nodes_per_ray"], "triangles_per_ray": k["tris_per_ray"], "avg_launch_ms": k["avg_ms"]} for name, k in kernels.items()},
            "shadow_skipping": {"state": ("undecided", "on", "off")[stats.shadow_skipping], "fixed_by_environment": bool(stats.shadow_skipping_fixed),
                                "decided_on_rays": int(stats.shadow_rays_decided_on),
                                "blocked_share": (stats.shadow_rays_blocked / stats.shadow_rays_decided_on) if stats.shadow_rays_decided_on else None},
            "bvh": {"nodes": info.node_count, "triangles": info.triangle_count, "depth": info.depth, "build_ms": info.build_ms},
            "cpu_baseline": cpu,
        }
    backend.close()
    return result


def main():
    args = parse_args()
    # The C host reports progress on stdout like the reference does; keep stdout for the one JSON line
    json_fd = os.dup(1)
    os.dup2(2, 1)
    result = run(args)
    if result is not None:
        import ctypes as C
        C.CDLL(None).fflush(None)
        os.write(json_fd, (json.dumps(result) + "\n").encode())


if __name__ == "__main__":
    main()
