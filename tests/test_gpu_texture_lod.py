"""Mip chains, trilinear filtering and the level-of-detail rule on the MI355X, through the C ABI, against the CPU
oracle bit for bit; plus what the rule is for: a minified high-frequency texture no longer aliases."""
import numpy as np
import pytest

import oracle
from path_tracer_amd import hip_api
from tests import synthetic
from tests.test_gpu_parity import bits, gpu_scene, same, to_gpu_constants
from tests.test_texture_lod_cpu import check_minification, minification_statistics, recovered_lod, textured_quad

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def soup():
    scene = synthetic.triangle_soup(seed=12, triangle_count=8000, material_count=5, texture_size=128)
    lights = synthetic.random_lights(4, 9)
    return scene, lights, oracle.OracleScene(scene), gpu_scene(scene, lights)


def random_hits(o_scene, seed, count):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-6, 6, (count, 3)).astype(np.float32)
    d = rng.normal(size=(count, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tri, tuv, front = o_scene.trace_rays(o, d)
    hit = tri >= 0
    return tri[hit], tuv[hit, 1:], front[hit], -d[hit]


def test_shading_data_exact_at_every_level_of_detail(soup):
    scene, lights, o_scene, g_scene = soup
    assert scene.textures[0].mip_count == 6
    tri, barys, front, out_dirs = random_hits(o_scene, 6, 60000)
    rng = np.random.default_rng(7)
    # From far below a texel to far above the whole texture, and a few exact zeros
    footprints = (10.0 ** rng.uniform(-5, 2, len(tri))).astype(np.float32)
    footprints[::97] = 0.0
    constants = oracle.write_constants(synthetic.look_at_spec((0, 9, 0), (np.pi / 2, 0, 0)), 64, 64, scene, lights)
    a = o_scene.shading_data(constants, tri, barys, front, out_dirs, footprints=footprints)
    g_scene.set_render_settings(3, 4, scene.emission_material_index)
    b = g_scene.evaluate_shading_data(to_gpu_constants(constants), tri, barys, front, out_dirs, footprints=footprints)
    assert same(a, b)
    # The footprint matters: most hits look different at level 0
    sharp = o_scene.shading_data(constants, tri, barys, front, out_dirs)
    assert (bits(sharp[:, 13]) != bits(a[:, 13])).mean() > 0.5


def test_level_of_detail_closed_form_on_the_gpu():
    size, edge = 64, 2.0
    scene, values = textured_quad(size, 1.0, edge)
    g_scene = gpu_scene(scene, np.zeros((0, 4), np.float32))
    constants = oracle.write_constants(synthetic.look_at_spec((1, 1, 3), (0, 0, 0)), 32, 32, scene, np.zeros((0, 4), np.float32))
    texel = edge / size
    footprints = (texel * 2.0 ** np.linspace(-2, 6, 33)).astype(np.float32)
    n = len(footprints)
    g_scene.set_render_settings(3, 4, 0)
    b = g_scene.evaluate_shading_data(to_gpu_constants(constants), np.zeros(n, np.int32), np.tile([0.3, 0.3], (n, 1)).astype(np.float32), np.ones(n, np.uint8),
                                      np.tile([0.0, 0.0, 1.0], (n, 1)).astype(np.float32), footprints=footprints)
    expected = np.clip(np.log2(footprints.astype(np.float64) / texel), 0, len(values) - 1)
    np.testing.assert_allclose(recovered_lod(b[:, 13], values), expected, atol=0.02)
    g_scene.close()


@pytest.mark.parametrize("strategy", [0, 1, 2, 3])
def test_render_exact_with_mip_chains(strategy, soup):
    scene, lights, o_scene, g_scene = soup
    width, height, frames = 72, 40, 3
    constants = oracle.write_constants(synthetic.look_at_spec((0, 9, 0), (np.pi / 2, 0, 0)), width, height, scene, lights)
    a, rays = o_scene.render(constants, lights, strategy=strategy, path_length=5, first_frame_index=31, frame_count=frames, threads=8)
    g_scene.set_render_settings(strategy, 5, scene.emission_material_index)
    g_scene.create_render_targets(width, height)
    g_scene.reset_statistics()
    g_scene.render_samples(to_gpu_constants(constants), 31, frames, accumulate=False)
    b, count = g_scene.read_accumulation()
    stats = g_scene.statistics()
    assert count == frames and stats.extension_rays + stats.shadow_rays == rays
    assert same(a, b)


def test_minified_texture_does_not_alias():
    """What the rule is for. Seen through a large footprint (a distant or indirectly seen surface), the lookups of
    a high-frequency texture with filtered mip levels return nearly the texture's mean instead of single texels:
    the per-sample spread of the albedo collapses while its mean stays."""
    scene = synthetic.triangle_soup(seed=5, triangle_count=2000, material_count=2, texture_size=128, filtered_mips=True)
    o_scene, g_scene = oracle.OracleScene(scene), gpu_scene(scene, np.zeros((0, 4), np.float32))
    hits = random_hits(o_scene, 9, 20000)
    constants = to_gpu_constants(oracle.write_constants(synthetic.look_at_spec((0, 9, 0), (np.pi / 2, 0, 0)), 64, 64, scene, np.zeros((0, 4), np.float32)))
    g_scene.set_render_settings(3, 4, scene.emission_material_index)
    check_minification(minification_statistics(scene, lambda t, b, f, o, fp: g_scene.evaluate_shading_data(constants, t, b, f, o, footprints=fp), hits))
    g_scene.close()
