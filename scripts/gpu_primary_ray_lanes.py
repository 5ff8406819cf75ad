"""How well the camera rays, the most coherent rays there are, already fill the lanes of the traversal kernel:
a render at path length 1 traces nothing else. Prints the kernel's own diagnostics (PT_TRACE_DIAGNOSTICS: lanes that
hold a ray, lanes that take part in a node step and in a triangle test) and the time per launch, then the same for
the full path length for comparison. A traversal that walks one tile of 8x8 camera rays together can at best bring
every one of those shares to 100 % - if the nodes that any ray of the tile visits were no more than one ray visits.
usage: python scripts/gpu_primary_ray_lanes.py [city|cornell|interior]"""
import os
import sys

os.environ["PT_TRACE_DIAGNOSTICS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from path_tracer_amd import hip_api, host_api  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "city"
args = bench.parse_args(["--workload", workload])
root = bench.workload_root(args, 0)
name, _, save, label, _ = bench.WORKLOADS[workload]
spec = host_api.quickload(os.path.join(root, "data", "saves", save))
for path_length in (1, 4):
    renderer = host_api.Renderer(spec, root, device=0, width=1920, height=1080, sampling_strategy=hip_api.SAMPLING_STRATEGY_NEE, path_length=path_length)
    renderer.render_frames(32)  # warm-up; decides about shadow-ray skipping
    renderer.synchronize()
    print("---- path length %d: visit counters" % path_length, flush=True)
    renderer.set_instrumentation(kernel_timing=False, visit_counters=True)
    renderer.reset_statistics()
    renderer.render_frames(32)
    s = renderer.statistics()
    print("extension rays %d, nodes per ray %.2f, triangles per ray %.2f" % (s.extension_rays, s.nodes_visited[0] / s.extension_rays, s.triangles_tested[0] / s.extension_rays), flush=True)
    renderer.set_instrumentation(kernel_timing=True, visit_counters=False)
    renderer.reset_statistics()
    renderer.render_frames(32)
    s = renderer.statistics()
    print("extension kernel: %.3f ms for %d rays in %d launches = %.2f ps per node visit at the node counts above" % (
        s.kernel_ms[1], s.extension_rays, s.kernel_launches[1], 0.0), flush=True)
    print("kernel ms:", [round(v, 3) for v in s.kernel_ms], flush=True)
    renderer.close()
