"""Differential test over randomly drawn configurations: scene size, light count, sampling strategy, light sampling
mode, path length, camera type, resolution (not a multiple of the 8x8 tiles), frame index and stride. Every image
must equal the oracle's bit for bit, and the ray counts must agree."""
import numpy as np
import pytest

import oracle
from path_tracer_amd import hip_api
from tests import synthetic
from tests.test_gpu_parity import gpu_scene, same, to_gpu_constants

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(12))
def test_random_configurations(seed):
    rng = np.random.default_rng(1000 + seed)
    triangle_count = int(rng.choice([1, 7, 300, 5000, 60000]))
    scene = synthetic.triangle_soup(seed=2000 + seed, triangle_count=triangle_count, material_count=int(rng.integers(1, 6)),
                                    texture_size=int(rng.choice([4, 16, 64])), emission_fraction=float(rng.choice([0.0, 0.05, 0.5])))
    o_scene = oracle.OracleScene(scene)
    g_scene = gpu_scene(scene, np.zeros((0, 4), np.float32))
    for trial in range(5):
        light_count = int(rng.choice([0, 1, 3, 32, 33, 90]))
        lights = synthetic.random_lights(int(rng.integers(1 << 30)), light_count)
        g_scene.set_lights(lights)
        strategy = int(rng.choice([0, 1, 2, 3, 3, 3]))
        light_sampling = int(rng.integers(0, 2)) if strategy == 3 else 0
        path_length = int(rng.choice([0, 1, 2, 4, 7]))
        width, height = int(rng.integers(1, 90)), int(rng.integers(1, 70))
        frames, stride, first = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(0, 1 << 20))
        position = rng.uniform(-7, 7, 3)
        spec = synthetic.look_at_spec(tuple(position), (float(rng.uniform(0.2, 2.9)), 0.0, float(rng.uniform(-3, 3))),
                                      camera_type=int(rng.integers(0, 4)), sky_strength=float(rng.choice([0.0, 1.0])))
        constants = oracle.write_constants(spec, width, height, scene, lights)
        expected, rays = o_scene.render(constants, lights, strategy=strategy, path_length=path_length, first_frame_index=first,
                                        frame_count=frames, frame_stride=stride, threads=16, light_sampling=light_sampling)
        g_scene.set_light_sampling(light_sampling)
        g_scene.create_render_targets(width, height)
        g_scene.set_render_settings(strategy, path_length, scene.emission_material_index)
        g_scene.reset_statistics()
        g_scene.render_samples(to_gpu_constants(constants), first, frames, stride, accumulate=False)
        got, count = g_scene.read_accumulation()
        stats = g_scene.statistics()
        label = dict(seed=seed, trial=trial, triangles=triangle_count, lights=light_count, strategy=strategy, tree=light_sampling,
                     path_length=path_length, size=(width, height), frames=frames, stride=stride, first=first)
        assert count == frames, label
        assert stats.extension_rays + stats.shadow_rays == rays, label
        assert same(got, expected), label
    g_scene.close()
