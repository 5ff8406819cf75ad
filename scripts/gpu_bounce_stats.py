"""Rays, shaded hits and shadow rays per path vertex of a benchmark workload, from runs at path lengths 1..4
(queue sizes are deterministic given the frame indices, so differences between path lengths are per-vertex counts).
usage: python scripts/gpu_bounce_stats.py [city|cornell|interior]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from path_tracer_amd import hip_api, host_api  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "city"
sys.argv = [sys.argv[0], "--workload", workload]
args = bench.parse_args()
root = bench.workload_root(args, 0)
name, _, save, label, _ = bench.WORKLOADS[workload]
spec = host_api.quickload(os.path.join(root, "data", "saves", save))
previous = (0, 0, 0)
for path_length in (1, 2, 3, 4):
    renderer = host_api.Renderer(spec, root, device=0, width=1920, height=1080, sampling_strategy=hip_api.SAMPLING_STRATEGY_NEE, path_length=path_length)
    renderer.set_instrumentation(kernel_timing=False, visit_counters=True)
    renderer.reset_statistics()
    renderer.render_frames(4)
    renderer.synchronize()
    s = renderer.statistics()
    current = (s.extension_rays, s.shaded_hits, s.shadow_rays)
    paths = s.paths
    print("vertex %d: extension rays %.3f per path, hits %.3f, shadow rays (from the previous vertex) %.3f" % (
        path_length, (current[0] - previous[0]) / paths, (current[1] - previous[1]) / paths, (current[2] - previous[2]) / paths), file=sys.stderr)
    previous = current
    renderer.close()
