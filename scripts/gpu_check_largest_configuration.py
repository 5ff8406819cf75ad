"""The largest configuration of the benchmark list at its full size, with checks attached (too slow to generate for
the test suite, which runs a tenth of it in tests/test_gpu_large_scene.py): a generated city of a hundred million
triangles with ten thousand spherical lights at 3840x2160, next event estimation through the light tree.

Asserted: facts about the BVH, conservation of rays and paths, finite radiance with alpha one, independence of the
batch layout (bit for bit), and the traversal of the large BVH tied to the CPU oracle through a subset: the oracle is
built on every hundredth triangle (one million), the library must match it bit for bit on that subset, and where the
closest hit of a ray in the full scene is a triangle of the subset the subset scene must return the same hit, else a
hit that is not nearer. Then the benchmark's throughput figures for the configuration.
usage: python scripts/gpu_check_largest_configuration.py [triangles] (prints a log; exits nonzero if a check fails)"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402  the checker
from oracle import formats  # noqa: E402
from path_tracer_amd import hip_api, host_api  # noqa: E402
from path_tracer_amd.tools import procedural  # noqa: E402
from tests.test_gpu_parity import bits, gpu_scene  # noqa: E402

TRIANGLES = int(sys.argv[1]) if len(sys.argv) > 1 else 100000000
WIDTH, HEIGHT, SUBSET = 3840, 2160, 100
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
start = time.time()
phase = ["starting"]


def heartbeat():
    while True:
        time.sleep(30)
        print("[%5.0f s] still %s" % (time.time() - start, phase[0]), flush=True)


threading.Thread(target=heartbeat, daemon=True).start()


def log(text):
    print("[%5.0f s] %s" % (time.time() - start, text), flush=True)


def check(condition, text):
    log(("ok:   " if condition else "FAIL: ") + text)
    if not condition:
        sys.exit(1)


phase[0] = "generating the scene"
root = os.path.join(os.environ.get("TMPDIR", "/tmp"), "pt_scenes", "city_%d" % TRIANGLES)
procedural.ensure_scene("city", root, triangles=TRIANGLES, lights_file=os.path.join(DATA, "Bistro_outside.lights"), save_file=os.path.join(DATA, "saves", "bistro", "outside.rt_save"))
log("scene generated")
phase[0] = "parsing the scene with numpy"
data = os.path.join(root, "data")
scene = formats.load_scene(os.path.join(data, "Bistro_outside.vks"), os.path.join(data, "Bistro_textures"))
check(scene.triangle_count >= TRIANGLES, "%d triangles, %d materials" % (scene.triangle_count, len(scene.material_names)))
rng = procedural.Pcg2d(4)
low = scene.dequantization_summand.astype(np.float64)
high = low + scene.dequantization_factor.astype(np.float64) * float((1 << 21) - 1)
lights = np.zeros((10000, 4), np.float32)
lights[:, :3] = rng.uniform(low, high, (10000, 3))
lights[:, 3] = rng.uniform(0.1, 0.5, 10000)

phase[0] = "loading the scene through the C host and building the BVH"
spec = host_api.quickload(os.path.join(data, "saves", "bistro", "outside.rt_save"))
os.environ.pop("PT_BATCH_PATHS", None)
renderer = host_api.Renderer(spec, root, device=0, width=WIDTH, height=HEIGHT)
lib = hip_api.lib()
check(lib.pt_set_lights(renderer.device_scene, lights.ctypes.data, len(lights)) == 0, "10000 lights set")
renderer.set_light_sampling(hip_api.LIGHT_SAMPLING_TREE)
info = renderer.bvh_info()
log("BVH: %d nodes of %d bytes, %d triangles of %d bytes, depth %d, built in %.0f ms" % (info.node_count, info.node_bytes, info.triangle_count, info.triangle_bytes, info.depth, info.build_ms))
check(info.triangle_count == scene.triangle_count and scene.triangle_count / 8 < info.node_count < scene.triangle_count and 8 <= info.depth <= 40, "BVH facts")
check(info.node_count * info.node_bytes + info.triangle_count * info.triangle_bytes > (4 << 30), "nodes and triangles take more than 4 GiB")

phase[0] = "rendering"
images = []
for batch_paths in (None, "9000000"):
    if batch_paths:
        os.environ["PT_BATCH_PATHS"] = batch_paths  # one frame per batch instead of both in one
        renderer.resize(WIDTH, HEIGHT)
        renderer.targets.accum_frame_count = 0
        renderer.spec.frame_index = spec.frame_index
    renderer.set_instrumentation(kernel_timing=False, visit_counters=True)
    renderer.reset_statistics()
    renderer.render_frames(2)
    image, samples = renderer.read_accumulation()
    stats = renderer.statistics()
    paths = 2 * WIDTH * HEIGHT
    check(samples == 2 and stats.paths == paths, "2 frames, %d paths" % paths)
    check(paths <= stats.extension_rays <= 4 * paths and stats.shadow_rays <= stats.shaded_hits <= stats.extension_rays and stats.shadow_rays > 0.2 * paths,
          "ray conservation: %d extension rays, %d shaded hits, %d shadow rays" % (stats.extension_rays, stats.shaded_hits, stats.shadow_rays))
    check(stats.nodes_visited[0] > 5 * stats.extension_rays and stats.triangles_tested[1] > stats.shadow_rays,
          "visits: %.1f nodes and %.1f triangles per extension ray, %.1f and %.1f per shadow ray" % (
              stats.nodes_visited[0] / stats.extension_rays, stats.triangles_tested[0] / stats.extension_rays, stats.nodes_visited[1] / stats.shadow_rays, stats.triangles_tested[1] / stats.shadow_rays))
    check(bool(np.isfinite(image).all() and (image[..., 3] == 1.0).all() and image[..., :3].min() >= 0.0 and image[..., :3].mean() > 1e-3),
          "image finite, alpha one, mean radiance %.4f" % (image[..., :3].mean() / 2))
    images.append(image)
os.environ.pop("PT_BATCH_PATHS", None)
check(bool((bits(images[0]) == bits(images[1])).all()), "both frames in one batch and one frame per batch give the same bits")

phase[0] = "tracing rays through the full BVH"
count = 400000
origins = rng.uniform(low, high, (count, 3)).astype(np.float32)
directions = rng.normal(size=(count, 3))
directions[:, 2] *= 0.3  # mostly along the streets, where rays travel far
directions = (directions / np.linalg.norm(directions, axis=1, keepdims=True)).astype(np.float32)
full = (np.zeros(count, np.int32), np.zeros((count, 3), np.float32), np.zeros(count, np.uint8))
check(lib.pt_trace_rays(renderer.device_scene, origins.ctypes.data, directions.ctypes.data, count, full[0].ctypes.data, full[1].ctypes.data, full[2].ctypes.data) == 0, "%d rays traced through the full BVH" % count)
phase[0] = "building and tracing the subset with the oracle"
subset = np.arange(0, scene.triangle_count, SUBSET)
vertex_rows = (subset[:, None] * 3 + np.arange(3)).reshape(-1)
small = formats.Scene(scene.material_names, len(subset), scene.dequantization_factor, scene.dequantization_summand,
                      scene.positions[vertex_rows], scene.normals_uvs[vertex_rows], scene.materials[subset], scene.textures)
expected = oracle.OracleScene(small).trace_rays(origins, directions)
phase[0] = "tracing the subset with the library"
g_small = gpu_scene(small, lights[:1])
got_small = g_small.trace_rays(origins, directions)
g_small.close()
check(bool((expected[0] == got_small[0]).all() and (bits(expected[1]) == bits(got_small[1])).all() and (expected[2] == got_small[2]).all()),
      "library and oracle agree bit for bit on every %dth triangle (%d triangles)" % (SUBSET, len(subset)))
hit_full, hit_small = full[0] >= 0, expected[0] >= 0
check(hit_full.mean() > 0.5 and bool(hit_full[hit_small].all()) and bool((full[1][hit_small, 0] <= expected[1][hit_small, 0]).all()),
      "%.1f %% of the rays hit the full scene; a hit in the subset is a hit in the full scene at that distance or nearer" % (100 * hit_full.mean()))
in_subset = hit_full & (full[0] % SUBSET == 0)
check(in_subset.sum() > 1000 and bool((expected[0][in_subset] == full[0][in_subset] // SUBSET).all())
      and bool((bits(expected[1][in_subset]) == bits(full[1][in_subset])).all()) and bool((expected[2][in_subset] == full[2][in_subset]).all()),
      "%d rays whose closest triangle in the full scene belongs to the subset: same hit bit for bit in the subset scene" % in_subset.sum())
others = hit_full & ~in_subset & hit_small
check(bool((expected[1][others, 0] >= full[1][others, 0]).all()), "elsewhere the subset's hit is not nearer (%d rays)" % others.sum())

phase[0] = "timing"
renderer.set_instrumentation(kernel_timing=True, visit_counters=False)
renderer.render_frames(16)
renderer.synchronize()
renderer.reset_statistics()
begin = time.perf_counter()
steps = 3
for _ in range(steps):
    renderer.render_frames(16)
renderer.synchronize()
seconds = time.perf_counter() - begin
stats = renderer.statistics()
rays = stats.extension_rays + stats.shadow_rays
log("throughput: %.0f Mrays/s, %.3f G samples/s, %.1f ms per 16-spp step; kernel ms per step: %s" % (
    rays / seconds * 1e-6, stats.paths / seconds * 1e-9, seconds / steps * 1e3, {n: round(stats.kernel_ms[i] / steps, 1) for i, n in enumerate(hip_api.KERNEL_NAMES)}))
renderer.close()
log("all checks passed")
