"""The benchmark's configurations at their full scene size and resolution, first frames only: the library against the
CPU oracle bit for bit (accumulation target and ray count). The driver-run suite does this at 256x144 on scenes a
seventh of the size; this run is the same comparison where the numbers of BASELINE.md are measured. The largest
configuration (100 M triangles) has scripts/gpu_check_largest_configuration.py instead: the oracle's builder does not
get through it in minutes.
usage (on a GPU box): python scripts/gpu_full_size_parity.py > gpurun_out/full_size_parity.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import oracle  # noqa: E402  the checker
from oracle import formats  # noqa: E402
from path_tracer_amd import host_api  # noqa: E402

# label, workload, width, height, frames
CASES = [("C1 Cornell box 512x512, 64 frames", "cornell", 512, 512, 64),
         ("C2 Cornell box 1920x1080, 8 frames", "cornell", 1920, 1080, 8),
         ("C3 generated city (2.8 M triangles, 28 lights) 1920x1080, 2 frames", "city", 1920, 1080, 2),
         ("C4 generated hall (1 M triangles, 32 lights) 3840x2160, 1 frame", "interior", 3840, 2160, 1)]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def main():
    failures = 0
    threads = bench.host_threads()
    for label, workload, width, height, frames in CASES:
        begin = time.time()
        args = bench.parse_args(["--workload", workload])
        root = bench.workload_root(args, 0)
        name, textures, save, _, _ = bench.WORKLOADS[workload]
        data = os.path.join(root, "data")
        scene = formats.load_scene(os.path.join(data, name + ".vks"), os.path.join(data, textures))
        lights = formats.load_lights(os.path.join(data, name + ".lights"))
        spec = formats.load_rt_save(os.path.join(data, "saves", save))
        renderer = host_api.Renderer(host_api.quickload(os.path.join(data, "saves", save)), root, device=0, width=width, height=height)
        constants = oracle.write_constants(spec, width, height, scene, lights)
        same_constants = bytes(renderer.constants()) == bytes(constants)
        renderer.reset_statistics()
        renderer.render_frames(frames)
        got, count = renderer.read_accumulation()
        stats = renderer.statistics()
        renderer.close()
        first = int(spec["frame_index"][0]) + 1
        cpu_begin = time.time()
        expected, rays = oracle.OracleScene(scene).render(constants, lights, first_frame_index=first, frame_count=frames, threads=threads)
        cpu_seconds = time.time() - cpu_begin
        differing = int((bits(got) != bits(expected)).any(axis=2).sum())
        ok = same_constants and count == frames and differing == 0 and stats.extension_rays + stats.shadow_rays == rays and bool(np.isfinite(got).all())
        failures += not ok
        print("%s: %s. %d triangles, frames %d..%d, %d paths, %d rays (oracle %d), %d of %d pixels differ, constants %s, "
              "mean radiance %.4f, oracle %.1f s on %d threads, case %.0f s"
              % ("ok  " if ok else "FAIL", label, scene.triangle_count, first, first + frames - 1, stats.paths, stats.extension_rays + stats.shadow_rays, rays,
                 differing, width * height, "equal" if same_constants else "DIFFER", float(got[..., :3].mean()) / frames, cpu_seconds, threads, time.time() - begin), flush=True)
    print("all cases bit-exact" if failures == 0 else "%d cases failed" % failures)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
