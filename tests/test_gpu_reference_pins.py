"""Files written by the reference's own texture converter (tests/golden/ref/, see tests/test_reference_pins.py)
through the HIP library: decoded once per scene by k_decode_texture, sampled by the shading path, compared with the
oracle bit for bit and with the levels the same converter stored uncompressed."""
import numpy as np
import pytest

import oracle
from oracle import formats
from tests.test_gpu_parity import gpu_scene, same, to_gpu_constants
from tests.test_reference_pins import check_reference_textures_through_shading, reference_textured_quad

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["blobs", "noise"])
def test_reference_converter_textures_on_the_gpu(name):
    scene, tri, barys, srgb_level, unorm_level = reference_textured_quad(name)
    g_scene = gpu_scene(scene, np.zeros((0, 4), np.float32))
    g_scene.set_render_settings(3, 4, 0)
    o_scene = oracle.OracleScene(scene)
    constants = oracle.write_constants(formats.default_scene_spec(), 32, 32, scene, np.zeros((0, 4), np.float32))
    gpu = lambda t, b, f, o: g_scene.evaluate_shading_data(to_gpu_constants(constants), t, b, f, o)
    if name == "blobs":
        out = check_reference_textures_through_shading(gpu, scene, tri, barys, srgb_level, unorm_level)
    else:
        n = len(tri)
        out = gpu(tri, barys, np.ones(n, np.uint8), np.tile([0.0, 0.0, 1.0], (n, 1)).astype(np.float32))
    n = len(tri)
    expected = o_scene.shading_data(constants, tri, barys, np.ones(n, np.uint8), np.tile([0.0, 0.0, 1.0], (n, 1)).astype(np.float32))
    assert same(expected, out)
    # every level of the file, at footprints from a texel to the whole square
    footprints = np.geomspace(2.0 / 64, 4.0, n).astype(np.float32)
    a = o_scene.shading_data(constants, tri, barys, np.ones(n, np.uint8), np.tile([0.0, 0.0, 1.0], (n, 1)).astype(np.float32), footprints=footprints)
    b = g_scene.evaluate_shading_data(to_gpu_constants(constants), tri, barys, np.ones(n, np.uint8), np.tile([0.0, 0.0, 1.0], (n, 1)).astype(np.float32), footprints=footprints)
    assert same(a, b)
    g_scene.close()
