"""Next event estimation through the light tree (pt_set_light_sampling): the library against the oracle's restatement
of the same definition, bit for bit, and against the reference's loop over all lights in the limit of many samples.
The reference itself has no light tree (it stops at 32 lights), so no fixture of the reference pins these images."""
import time

import numpy as np
import pytest

import oracle
from path_tracer_amd import hip_api
from tests import synthetic
from tests.test_gpu_parity import gpu_scene, same, to_gpu_constants

pytestmark = pytest.mark.gpu


def render(g_scene, constants, scene, first, frames, width, height, path_length=4):
    g_scene.create_render_targets(width, height)
    g_scene.set_render_settings(3, path_length, scene.emission_material_index)
    g_scene.reset_statistics()
    g_scene.render_samples(to_gpu_constants(constants), first, frames, 1, accumulate=False)
    got, count = g_scene.read_accumulation()
    assert count == frames
    stats = g_scene.statistics()
    return got, stats.extension_rays + stats.shadow_rays


@pytest.mark.parametrize("light_count", [1, 33, 700])
def test_light_tree_exact_soup(light_count):
    scene = synthetic.triangle_soup(seed=50 + light_count, triangle_count=8000, material_count=4, texture_size=32)
    lights = synthetic.random_lights(60 + light_count, light_count)
    g_scene = gpu_scene(scene, lights)
    g_scene.set_light_sampling(hip_api.LIGHT_SAMPLING_TREE)
    spec = synthetic.look_at_spec((0.5, 7.0, 0.3), (1.3, 0.0, 0.4))
    constants = oracle.write_constants(spec, 96, 64, scene, lights)
    got, rays = render(g_scene, constants, scene, 11, 2, 96, 64)
    o_scene = oracle.OracleScene(scene)
    expected, expected_rays = o_scene.render(constants, lights, strategy=3, path_length=4, first_frame_index=11, frame_count=2, threads=16, light_sampling=1)
    assert rays == expected_rays
    assert same(got, expected)
    # Back to the loop over all lights: the reference's image again
    g_scene.set_light_sampling(hip_api.LIGHT_SAMPLING_EXACT)
    got, rays = render(g_scene, constants, scene, 11, 2, 96, 64)
    expected, expected_rays = o_scene.render(constants, lights, strategy=3, path_length=4, first_frame_index=11, frame_count=2, threads=16)
    assert rays == expected_rays and same(got, expected)
    g_scene.close()


def test_light_tree_exact_room_and_replaced_lights():
    """Lights inscribed with emitters; replacing the light list rebuilds the tree."""
    scene, lights = synthetic.lit_room(seed=5, light_count=150)
    g_scene = gpu_scene(scene, lights)
    g_scene.set_light_sampling(hip_api.LIGHT_SAMPLING_TREE)
    o_scene = oracle.OracleScene(scene)
    spec = synthetic.look_at_spec((4.0, -5.0, 3.0), (1.15, 0.0, 2.5), sky_strength=0.0)
    for subset in (lights, lights[:40], lights[::-1]):
        g_scene.set_lights(subset)
        constants = oracle.write_constants(spec, 128, 80, scene, subset)
        got, rays = render(g_scene, constants, scene, 3, 2, 128, 80)
        expected, expected_rays = o_scene.render(constants, subset, strategy=3, path_length=4, first_frame_index=3, frame_count=2, threads=16, light_sampling=1)
        assert rays == expected_rays and same(got, expected)
        assert got[..., :3].mean() > 0.01
    g_scene.close()


@pytest.mark.parametrize("layout", synthetic.LIGHT_LAYOUTS)
def test_light_tree_exact_for_light_layouts(layout):
    """Light sets that stress the grid over the lights and its fallback to the walk of the tree (more than four lights
    met, crowded cells, a light listed in many cells, shading points inside a light, one layer of cells, shading points
    beyond the grid's reach): the library and the oracle agree bit for bit."""
    lights = synthetic.light_layout(layout)
    scene = synthetic.triangle_soup(seed=31, triangle_count=1200, material_count=3)
    g_scene = gpu_scene(scene, lights)
    g_scene.set_light_sampling(hip_api.LIGHT_SAMPLING_TREE)
    spec = synthetic.look_at_spec((0.5, 7.0, 0.3), (1.3, 0.0, 0.4))
    constants = oracle.write_constants(spec, 96, 64, scene, lights)
    got, rays = render(g_scene, constants, scene, 3, 2, 96, 64)
    expected, expected_rays = oracle.OracleScene(scene).render(constants, lights, strategy=3, path_length=4, first_frame_index=3, frame_count=2, threads=16, light_sampling=1)
    assert rays == expected_rays
    assert same(got, expected)
    assert got[..., :3].sum() > 0
    g_scene.close()


def test_light_sampling_rejects_unknown_mode():
    scene = synthetic.triangle_soup(seed=1, triangle_count=10, material_count=2, texture_size=4)
    g_scene = gpu_scene(scene, np.zeros((0, 4), np.float32))
    assert hip_api.lib().pt_set_light_sampling(g_scene.handle, 2) != 0
    assert hip_api.lib().pt_set_light_sampling(None, 1) != 0
    # No lights at all: both modes render the unlit scene
    spec = synthetic.look_at_spec((0.5, 7.0, 0.3), (1.3, 0.0, 0.4))
    constants = oracle.write_constants(spec, 32, 32, scene, np.zeros((0, 4), np.float32))
    g_scene.set_light_sampling(hip_api.LIGHT_SAMPLING_TREE)
    tree, _ = render(g_scene, constants, scene, 1, 1, 32, 32)
    g_scene.set_light_sampling(hip_api.LIGHT_SAMPLING_EXACT)
    loop, _ = render(g_scene, constants, scene, 1, 1, 32, 32)
    assert same(tree, loop)
    g_scene.close()


def test_light_tree_converges_and_is_faster_with_ten_thousand_lights():
    """10 000 lights (the count of the largest configuration), direct light on the floor of the lit room: both
    estimators converge to the same image, and the tree's shading time no longer grows with the light count."""
    scene, lights = synthetic.lit_room(seed=4, light_count=10000, extent=120.0)
    g_scene = gpu_scene(scene, lights)
    spec = synthetic.look_at_spec((30.0, -40.0, 0.6), (0.6, 0.0, 2.5), sky_strength=0.0)
    width, height, spp = 160, 96, 256
    constants = oracle.write_constants(spec, width, height, scene, lights)
    g_scene.set_instrumentation(kernel_timing=True)
    images, shade_ms = {}, {}
    for name, mode, first in (("tree", hip_api.LIGHT_SAMPLING_TREE, 1), ("loop", hip_api.LIGHT_SAMPLING_EXACT, 1 + spp)):
        g_scene.set_light_sampling(mode)
        render(g_scene, constants, scene, first, 1, width, height, path_length=2)  # the first launch of a kernel loads its code
        begin = time.perf_counter()
        got, _ = render(g_scene, constants, scene, first, spp, width, height, path_length=2)
        shade_ms[name] = g_scene.statistics().kernel_ms[3]
        images[name] = got[..., :3] / spp
        print(name, "%.2f s, shading %.1f ms" % (time.perf_counter() - begin, shade_ms[name]))
    assert np.isfinite(images["tree"]).all() and np.isfinite(images["loop"]).all()
    mean = images["loop"].mean()
    assert mean > 0.01
    assert abs(images["tree"].mean() - mean) / mean < 0.01
    rmse = np.sqrt(((images["tree"] - images["loop"]) ** 2).mean()) / mean
    print("relative RMSE between the estimators at %d spp: %.3f" % (spp, rmse))
    assert rmse < 0.25
    assert shade_ms["tree"] * 10 < shade_ms["loop"]
    g_scene.close()
