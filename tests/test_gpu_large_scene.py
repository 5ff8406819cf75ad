"""The largest configuration of the benchmark list in the driver-run suite, as far as one test may take: a generated
city of ten million triangles with ten thousand spherical lights at 3840x2160, next event estimation through the
light tree. (The configuration names about a hundred million triangles; generating those takes minutes of CPU
time per run, a tenth of them exercises the same regime: a BVH of gigabytes that no cache holds, a tree of lights,
8.3 million paths per frame.)

Checked: facts about the BVH, conservation of rays and paths, finite radiance with alpha one, independence of the
batch layout, and the traversal of the large BVH against the oracle. The oracle's median-split builder would take
about a minute on ten million triangles, so it is built on every tenth triangle (one million), and the large BVH
is tied to it through the rays whose closest hit in the full scene is a triangle of that subset: there the subset
scene must return the same hit bit for bit, and everywhere its hit can only be as far or farther."""
import os

import numpy as np
import pytest

import oracle
from oracle import formats
from path_tracer_amd import hip_api, host_api
from path_tracer_amd.tools import procedural
from tests.paths import DATA
from tests.test_gpu_parity import bits, gpu_scene

pytestmark = pytest.mark.gpu

WIDTH, HEIGHT = 3840, 2160


def generated_lights(count, scene):
    """count lights uniform in the scene's bounding box, radius 0.1 to 0.5 (the benchmark's rule, seed 4)."""
    rng = np.random.default_rng(4)
    low = scene.dequantization_summand.astype(np.float64)
    high = low + scene.dequantization_factor.astype(np.float64) * float((1 << 21) - 1)
    lights = np.zeros((count, 4), np.float32)
    lights[:, :3] = rng.uniform(low, high, (count, 3))
    lights[:, 3] = rng.uniform(0.1, 0.5, count)
    return lights


@pytest.fixture(scope="module")
def large_scene(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("large"))
    procedural.ensure_scene("city", root, triangles=10000000, texture_size=64, lights_file=os.path.join(DATA, "Bistro_outside.lights"),
                            save_file=os.path.join(DATA, "saves", "bistro", "outside.rt_save"))
    data = os.path.join(root, "data")
    scene = formats.load_scene(os.path.join(data, "Bistro_outside.vks"), os.path.join(data, "Bistro_textures"))
    assert scene.triangle_count >= 10000000
    return root, scene, generated_lights(10000, scene)


def test_render_4k_with_ten_thousand_lights(large_scene, monkeypatch):
    root, scene, lights = large_scene
    spec = host_api.quickload(os.path.join(root, "data", "saves", "bistro", "outside.rt_save"))
    images = []
    for batch_paths in (None, "9000000"):
        if batch_paths is None:
            monkeypatch.delenv("PT_BATCH_PATHS", raising=False)
        else:
            # 8.3 M paths per frame: one frame per batch instead of both in one
            monkeypatch.setenv("PT_BATCH_PATHS", batch_paths)
        renderer = host_api.Renderer(spec, root, device=0, width=WIDTH, height=HEIGHT)
        assert hip_api.lib().pt_set_lights(renderer.device_scene, lights.ctypes.data, len(lights)) == 0
        renderer.set_light_sampling(hip_api.LIGHT_SAMPLING_TREE)
        info = renderer.bvh_info()
        assert info.triangle_count == scene.triangle_count
        # Nodes and triangles together are beyond every cache of the part (256 MiB at most)
        assert info.node_count * info.node_bytes + info.triangle_count * info.triangle_bytes > (512 << 20) and 8 <= info.depth <= 40
        assert scene.triangle_count / 8 < info.node_count < scene.triangle_count
        renderer.set_instrumentation(kernel_timing=False, visit_counters=True)
        renderer.reset_statistics()
        renderer.render_frames(2)
        image, samples = renderer.read_accumulation()
        stats = renderer.statistics()
        paths = 2 * WIDTH * HEIGHT
        assert samples == 2 and stats.paths == paths
        # Every path has its primary ray; a path of length 4 has at most 4 extension and 3 shadow rays, and
        # shadow rays leave shaded hits only
        assert paths <= stats.extension_rays <= 4 * paths
        assert stats.shadow_rays <= stats.shaded_hits <= stats.extension_rays
        assert stats.shadow_rays > 0.2 * paths
        assert stats.nodes_visited[0] > 5 * stats.extension_rays and stats.triangles_tested[1] > stats.shadow_rays
        assert np.isfinite(image).all() and (image[..., 3] == 1.0).all()
        assert image[..., :3].min() >= 0.0 and image[..., :3].mean() > 1e-3
        images.append(image)
        renderer.close()
    assert (bits(images[0]) == bits(images[1])).all()


def test_traversal_of_the_large_bvh_against_the_oracle(large_scene):
    root, scene, lights = large_scene
    subset = np.arange(0, scene.triangle_count, 10)
    vertex_rows = (subset[:, None] * 3 + np.arange(3)).reshape(-1)
    small = formats.Scene(scene.material_names, len(subset), scene.dequantization_factor, scene.dequantization_summand,
                          scene.positions[vertex_rows], scene.normals_uvs[vertex_rows], scene.materials[subset], scene.textures)
    rng = np.random.default_rng(17)
    count = 100000
    low = scene.dequantization_summand.astype(np.float64)
    high = low + scene.dequantization_factor.astype(np.float64) * float((1 << 21) - 1)
    origins = rng.uniform(low, high, (count, 3)).astype(np.float32)
    directions = rng.normal(size=(count, 3))
    directions[:, 2] *= 0.3  # mostly along the streets, where rays travel far
    directions = (directions / np.linalg.norm(directions, axis=1, keepdims=True)).astype(np.float32)
    # The oracle on the subset, and the library on the subset: bit-exact
    expected = oracle.OracleScene(small).trace_rays(origins, directions)
    g_small = gpu_scene(small, lights[:1])
    got_small = g_small.trace_rays(origins, directions)
    g_small.close()
    assert (expected[0] == got_small[0]).all() and (bits(expected[1]) == bits(got_small[1])).all() and (expected[2] == got_small[2]).all()
    # The library on all ten million triangles
    g_full = gpu_scene(scene, lights[:1])
    full = g_full.trace_rays(origins, directions)
    g_full.close()
    hit_full, hit_small = full[0] >= 0, expected[0] >= 0
    assert hit_full.mean() > 0.5
    # A hit in the subset is a hit in the full scene, at that distance or nearer
    assert hit_full[hit_small].all()
    assert (full[1][hit_small, 0] <= expected[1][hit_small, 0]).all()
    # Where the full scene's closest triangle is one of the subset, the subset scene finds the same hit
    in_subset = hit_full & (full[0] % 10 == 0)
    assert in_subset.sum() > 2000
    assert (expected[0][in_subset] == full[0][in_subset] // 10).all()
    assert (bits(expected[1][in_subset]) == bits(full[1][in_subset])).all() and (expected[2][in_subset] == full[2][in_subset]).all()
    # and where it is not, the subset scene's hit is strictly another triangle, not nearer
    others = hit_full & ~in_subset & hit_small
    assert (expected[1][others, 0] >= full[1][others, 0]).all()
