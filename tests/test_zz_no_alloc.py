"""pt_trace_rays_no_alloc(): the entry point that allocates nothing, checked through the C ABI against
pt_trace_rays() and against the CPU oracle bit for bit, twice in a row with the second call on fewer rays (the
scratch array keeps what the first call left in it)."""
import ctypes as C

import numpy as np
import pytest
import torch  # before the HIP library is loaded, so that both share one HIP runtime in this process

import oracle
from path_tracer_amd import hip_api
from tests import synthetic

pytestmark = pytest.mark.gpu


def device_scene(scene):
    textures = [(t.width, t.height, t.format, [t.payload[int(t.mip_offsets[m]):int(t.mip_offsets[m]) + t.mip_extents[m][2]] for m in range(t.mip_count)])
                for t in scene.textures]
    return hip_api.Scene(len(scene.material_names), scene.dequantization_factor, scene.dequantization_summand,
                         scene.positions, scene.normals_uvs, scene.materials, textures)


def random_rays(seed, count, extent):
    rng = np.random.default_rng(seed)
    origins = rng.uniform(-extent, extent, (count, 3)).astype(np.float32)
    directions = rng.normal(size=(count, 3))
    directions = (directions / np.linalg.norm(directions, axis=1, keepdims=True)).astype(np.float32)
    return origins, directions


def trace_no_alloc(scene, origins, directions):
    """Inputs and outputs in device memory of the caller (torch tensors here), as the entry point asks."""
    count = len(origins)
    d_origins = torch.from_numpy(origins).cuda()
    d_directions = torch.from_numpy(directions).cuda()
    d_triangles = torch.full((count,), -7, dtype=torch.int32, device="cuda")
    d_tuv = torch.full((count, 3), -7.0, dtype=torch.float32, device="cuda")
    d_front = torch.full((count,), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    code = hip_api.lib().pt_trace_rays_no_alloc(scene.handle, d_origins.data_ptr(), d_directions.data_ptr(), count,
                                                d_triangles.data_ptr(), d_tuv.data_ptr(), d_front.data_ptr())
    assert code == 0
    return d_triangles.cpu().numpy(), d_tuv.cpu().numpy(), d_front.cpu().numpy()


def test_no_alloc_matches_trace_rays_twice_in_a_row():
    soup = synthetic.triangle_soup(seed=21, triangle_count=20000, material_count=3)
    scene = device_scene(soup)
    checker = oracle.OracleScene(soup)
    try:
        for seed, count in ((1, 150000), (2, 4321)):
            origins, directions = random_rays(seed, count, 6.0)
            expected = scene.trace_rays(origins, directions)
            got = trace_no_alloc(scene, origins, directions)
            assert (expected[0] >= 0).sum() > count // 10
            for reference in (expected, checker.trace_rays(origins, directions)):
                assert np.array_equal(reference[0], got[0])
                assert np.array_equal(reference[1].view(np.uint32), got[1].view(np.uint32))
                assert np.array_equal(reference[2], got[2])
    finally:
        scene.close()


def test_no_alloc_refuses_too_many_rays():
    scene = device_scene(synthetic.triangle_soup(seed=22, triangle_count=100, material_count=2))
    try:
        dummy = torch.zeros(16, device="cuda")
        code = hip_api.lib().pt_trace_rays_no_alloc(scene.handle, dummy.data_ptr(), dummy.data_ptr(), (1 << 20) + 1,
                                                    dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr())
        assert code != 0
    finally:
        scene.close()
