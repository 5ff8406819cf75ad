"""Parity of the HIP path (through the C ABI and the C host) with the CPU oracle on a real MI355X.

Bars: bit-exact for the integer RNG stream, for closest-hit triangle indices, barycentrics and front flags and,
because both sides use the same order of IEEE operations, for the accumulated fp32 radiance; where float sums are
reordered (multi-GPU style slices) the tolerance is 1e-5 relative to the pixel sum.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from oracle import formats
from path_tracer_amd import hip_api, host_api
from tests import synthetic
from tests.paths import DATA, DATA_ROOT

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(a, b):
    return ((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all()


def to_gpu_constants(constants):
    return hip_api.Constants.from_buffer_copy(bytes(constants))


def gpu_scene(scene, lights):
    textures = [(t.width, t.height, t.format, [t.payload[int(t.mip_offsets[m]):int(t.mip_offsets[m]) + t.mip_extents[m][2]] for m in range(t.mip_count)])
                for t in scene.textures]
    s = hip_api.Scene(len(scene.material_names), scene.dequantization_factor, scene.dequantization_summand,
                      scene.positions, scene.normals_uvs, scene.materials, textures)
    s.set_lights(lights)
    return s


@pytest.fixture(scope="module")
def gpu_cornell(cornell, cornell_lights):
    return gpu_scene(cornell, cornell_lights)


@pytest.fixture(scope="module")
def soup():
    scene = synthetic.triangle_soup(seed=2, triangle_count=20000, material_count=5, texture_size=64)
    lights = synthetic.random_lights(3, 40)  # more than the reference's limit of 32
    return scene, lights, oracle.OracleScene(scene), gpu_scene(scene, lights)


def test_rng_stream_bit_exact():
    for x, y, f in ((0, 0, 0), (255, 255, 2696), (1919, 1079, 65535), (7, 3, 70000)):
        got = hip_api.random_numbers(x, y, f, 9)
        expected, _ = oracle.rng_stream(x, y, f, 9)
        assert (bits(got) == bits(expected)).all()
    assert hip_api.random_numbers(0, 0, 0, 1)[0, 0] == np.float32(0.5007143)


def test_bvh_info(gpu_cornell, soup):
    info = gpu_cornell.bvh_info()
    assert info.triangle_count == 1310 and info.node_bytes == 80 and info.triangle_bytes == 48
    assert 1310 / 24 <= info.node_count <= 1310 and 1 <= info.depth <= 12
    info = soup[3].bvh_info()
    assert info.triangle_count == 20000 and info.node_count < 20000


def test_traversal_exact(gpu_cornell, oracle_cornell, soup):
    rng = np.random.default_rng(1)
    for o_scene, g_scene, lo, hi in ((oracle_cornell, gpu_cornell, [0, -0.56, 0], [0.556, 0.4, 0.549]), (soup[2], soup[3], [-6, -6, -6], [6, 6, 6])):
        o = rng.uniform(lo, hi, (100000, 3)).astype(np.float32)
        d = rng.normal(size=(100000, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[:300] = np.eye(3, dtype=np.float32)[np.arange(300) % 3] * np.where(np.arange(300) % 2, 1, -1)[:, None]
        a, b = o_scene.trace_rays(o, d), g_scene.trace_rays(o, d)
        assert (a[0] == b[0]).all() and (bits(a[1]) == bits(b[1])).all() and (a[2] == b[2]).all()
        assert (a[0] >= 0).mean() > 0.3


@pytest.mark.parametrize("distance", [1.0, 100.0, 10000.0])
def test_traversal_exact_for_rays_from_far_away(distance):
    """Rays that start up to 10 000 scene extents away (the CPU suite checks the same rays against a loop over all
    triangles): the persistent traversal kernel agrees with the oracle's binary BVH bit for bit."""
    from tests.test_kernel_bodies_cpu import axis_aligned_terraces, far_rays
    scene = axis_aligned_terraces()
    g_scene = gpu_scene(scene, np.zeros((0, 4), np.float32))
    origins, directions = far_rays(np.random.default_rng(int(distance)), 60000, distance)
    a = oracle.OracleScene(scene).trace_rays(origins, directions)
    b = g_scene.trace_rays(origins, directions)
    g_scene.close()
    assert (a[0] == b[0]).all() and (bits(a[1]) == bits(b[1])).all() and (a[2] == b[2]).all()
    assert (a[0] >= 0).mean() > 0.6


def test_primary_hits_exact_512(gpu_cornell, oracle_cornell, cornell, cornell_lights, cornell_spec):
    """BASELINE config 1 geometry: 512x512 primary rays of frame 2696, triangle IDs bit-exact (no ties to document)."""
    constants = oracle.write_constants(cornell_spec, 512, 512, cornell, cornell_lights)
    gpu_cornell.create_render_targets(512, 512)
    a = oracle_cornell.primary_hits(constants, 2696)
    b = gpu_cornell.trace_primary_rays(to_gpu_constants(constants), 2696)
    assert (a[0] == b[0]).all() and (bits(a[1]) == bits(b[1])).all() and (a[2] == b[2]).all()
    assert (a[0] >= 0).mean() > 0.95


def test_shading_data_exact(soup):
    scene, lights, o_scene, g_scene = soup
    rng = np.random.default_rng(5)
    o = rng.uniform(-6, 6, (50000, 3)).astype(np.float32)
    d = rng.normal(size=(50000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tri, tuv, front = o_scene.trace_rays(o, d)
    hit = tri >= 0
    constants = oracle.write_constants(synthetic.look_at_spec((0, 9, 0), (np.pi / 2, 0, 0)), 64, 64, scene, lights)
    a = o_scene.shading_data(constants, tri[hit], tuv[hit, 1:], front[hit], -d[hit])
    g_scene.set_render_settings(3, 4, scene.emission_material_index)
    b = g_scene.evaluate_shading_data(to_gpu_constants(constants), tri[hit], tuv[hit, 1:], front[hit], -d[hit])
    assert same(a, b)
    assert len(np.unique(a[:, 13])) > 100


@pytest.mark.parametrize("strategy", [0, 1, 2, 3])
@pytest.mark.parametrize("path_length", [0, 1, 2, 4, 7])
def test_render_exact_cornell(strategy, path_length, gpu_cornell, oracle_cornell, cornell, cornell_lights, cornell_spec):
    width, height, frames = 100, 60, 5  # not multiples of the 8x8 tile
    constants = oracle.write_constants(cornell_spec, width, height, cornell, cornell_lights)
    expected, rays = oracle_cornell.render(constants, cornell_lights, strategy=strategy, path_length=path_length, first_frame_index=2696, frame_count=frames, threads=16)
    gpu_cornell.create_render_targets(width, height)
    gpu_cornell.set_render_settings(strategy, path_length, cornell.emission_material_index)
    gpu_cornell.reset_statistics()
    gpu_cornell.render_samples(to_gpu_constants(constants), 2696, frames, 1, accumulate=False)
    got, count = gpu_cornell.read_accumulation()
    stats = gpu_cornell.statistics()
    assert count == frames
    assert stats.extension_rays + stats.shadow_rays == rays
    assert stats.paths == width * height * frames
    assert same(got, expected)


def test_render_exact_config1_cornell_512(gpu_cornell, oracle_cornell, cornell, cornell_lights, cornell_spec):
    """BASELINE config 1 at reduced sample count: 512x512, NEE, path length 4, frames 2696.. (8 of the 64 spp)."""
    constants = oracle.write_constants(cornell_spec, 512, 512, cornell, cornell_lights)
    expected, rays = oracle_cornell.render(constants, cornell_lights, strategy=3, path_length=4, first_frame_index=2696, frame_count=8, threads=16)
    gpu_cornell.create_render_targets(512, 512)
    gpu_cornell.set_render_settings(3, 4, cornell.emission_material_index)
    gpu_cornell.reset_statistics()
    gpu_cornell.render_samples(to_gpu_constants(constants), 2696, 8, 1, accumulate=False)
    got, count = gpu_cornell.read_accumulation()
    stats = gpu_cornell.statistics()
    assert stats.extension_rays + stats.shadow_rays == rays
    assert same(got, expected)
    mean = got[..., :3] / count
    reference = expected[..., :3] / 8
    rmse = np.sqrt(np.mean((mean - reference) ** 2)) / np.mean(reference)
    assert rmse == 0.0


@pytest.mark.parametrize("camera_type", [0, 1, 2, 3])
def test_render_exact_soup_cameras_many_lights(camera_type, soup):
    scene, lights, o_scene, g_scene = soup
    spec = synthetic.look_at_spec((0.5, 7.0, 0.3), (1.3, 0.0, 0.4), camera_type=camera_type)
    constants = oracle.write_constants(spec, 96, 64, scene, lights)
    expected, rays = o_scene.render(constants, lights, strategy=3, path_length=5, first_frame_index=77, frame_count=3, frame_stride=2, threads=16)
    g_scene.create_render_targets(96, 64)
    g_scene.set_render_settings(3, 5, scene.emission_material_index)
    g_scene.reset_statistics()
    g_scene.render_samples(to_gpu_constants(constants), 77, 3, 2, accumulate=False)
    got, _ = g_scene.read_accumulation()
    stats = g_scene.statistics()
    assert stats.extension_rays + stats.shadow_rays == rays
    assert same(got, expected)
    assert np.nanmax(got[..., :3]) > 0


@pytest.mark.parametrize("mode", ["0", "1", None])
def test_shadow_rays_skip_what_cannot_matter(mode, monkeypatch):
    """Shadow rays only report whether their closest hit is emissive; while it is not, they skip nodes and triangles
    without emissive triangles below them, and walk the tree a second time if the closest hit turns out emissive after
    all. PT_SHADOW_SKIPPING fixes that on or off; otherwise the first batch decides from the share of blocked rays. The
    image is the oracle's in every case: a room whose lamps are emissive octahedra behind and in front of walls, and a
    soup in which any triangle may be emissive."""
    if mode is None:
        monkeypatch.delenv("PT_SHADOW_SKIPPING", raising=False)
    else:
        monkeypatch.setenv("PT_SHADOW_SKIPPING", mode)
    room, room_lights = synthetic.lit_room(seed=9, light_count=40)
    soup_scene = synthetic.triangle_soup(seed=77, triangle_count=6000, material_count=3, texture_size=16)
    for scene, lights, spec in ((room, room_lights, synthetic.look_at_spec((4.0, -5.0, 3.0), (1.15, 0.0, 2.5), sky_strength=0.0)),
                                (soup_scene, synthetic.random_lights(5, 12), synthetic.look_at_spec((0.5, 7.0, 0.3), (1.3, 0.0, 0.4)))):
        g_scene = gpu_scene(scene, lights)
        o_scene = oracle.OracleScene(scene)
        constants = oracle.write_constants(spec, 128, 72, scene, lights)
        g_scene.create_render_targets(128, 72)
        for emission_material in (scene.emission_material_index, 0, 250):
            g_scene.set_render_settings(3, 4, emission_material)
            # Two batches: the second one runs with what the first one decided
            for first in (5, 9):
                g_scene.render_samples(to_gpu_constants(constants), first, 2, 1, accumulate=False)
                got, _ = g_scene.read_accumulation()
                expected, _ = o_scene.render(constants, lights, strategy=3, path_length=4, first_frame_index=first, frame_count=2, threads=16,
                                             emission_material_index=emission_material)
                assert same(got, expected), (mode, emission_material, first)
        g_scene.close()


def test_accumulation_semantics(gpu_cornell, cornell, cornell_lights, cornell_spec):
    """Two calls that accumulate equal one call; a call without accumulate restarts (src/main.c:1884-1888)."""
    constants = to_gpu_constants(oracle.write_constants(cornell_spec, 128, 72, cornell, cornell_lights))
    gpu_cornell.create_render_targets(128, 72)
    gpu_cornell.set_render_settings(3, 4, cornell.emission_material_index)
    gpu_cornell.render_samples(constants, 10, 9, 1, accumulate=False)
    one, n1 = gpu_cornell.read_accumulation()
    gpu_cornell.render_samples(constants, 10, 4, 1, accumulate=False)
    gpu_cornell.render_samples(constants, 14, 5, 1, accumulate=True)
    two, n2 = gpu_cornell.read_accumulation()
    assert n1 == n2 == 9 and same(one, two)
    gpu_cornell.render_samples(constants, 10, 9, 1, accumulate=False)
    again, n3 = gpu_cornell.read_accumulation()
    assert n3 == 9 and same(one, again)
    assert (one[..., 3] == 1).all()


def test_sample_slices_1080p(gpu_cornell, cornell, cornell_lights, cornell_spec):
    """BASELINE config 2 geometry (1920x1080): interleaved slices as used across GPUs sum to the plain render."""
    constants = to_gpu_constants(oracle.write_constants(cornell_spec, 1920, 1080, cornell, cornell_lights))
    gpu_cornell.create_render_targets(1920, 1080)
    gpu_cornell.set_render_settings(3, 4, cornell.emission_material_index)
    gpu_cornell.reset_statistics()
    gpu_cornell.render_samples(constants, 2696, 8, 1, accumulate=False)
    full, _ = gpu_cornell.read_accumulation()
    rays_full = gpu_cornell.statistics().extension_rays + gpu_cornell.statistics().shadow_rays
    gpu_cornell.reset_statistics()
    total = np.zeros_like(full)
    for rank in range(4):
        gpu_cornell.render_samples(constants, 2696 + rank, 2, 4, accumulate=False)
        part, count = gpu_cornell.read_accumulation()
        assert count == 2
        total += part
    rays_parts = gpu_cornell.statistics().extension_rays + gpu_cornell.statistics().shadow_rays
    assert rays_parts == rays_full
    np.testing.assert_allclose(total[..., :3], full[..., :3], rtol=1e-5, atol=1e-6)
    assert np.isfinite(full).all() and full[..., :3].mean() / 8 > 0.1


def test_host_loader_path_matches_direct_upload(gpu_cornell, cornell, cornell_lights, cornell_spec, oracle_cornell):
    """The C host (file loaders, camera, constants, frame counting) drives the same result as arrays handed over directly."""
    spec = host_api.quickload(os.path.join(DATA, "saves", "cornell_box", "default.rt_save"))
    renderer = host_api.Renderer(spec, DATA_ROOT, device=0, width=160, height=90)
    assert renderer.lit_scene.scene.emission_material_index == 0 and renderer.lit_scene.spherical_light_count == 1
    expected_constants = oracle.write_constants(cornell_spec, 160, 90, cornell, cornell_lights)
    assert bytes(renderer.constants()) == bytes(expected_constants)
    renderer.render_frames(3)
    renderer.render_frames(2)
    assert renderer.spec.frame_index == 2700 and renderer.targets.accum_frame_count == 5
    got, count = renderer.read_accumulation()
    expected, _ = oracle_cornell.render(expected_constants, cornell_lights, first_frame_index=2696, frame_count=5, threads=16)
    assert count == 5 and same(got, expected)
    renderer.close()


def test_screenshots(tmp_path, cornell_spec):
    from PIL import Image
    spec = host_api.quickload(os.path.join(DATA, "saves", "cornell_box", "default.rt_save"))
    renderer = host_api.Renderer(spec, DATA_ROOT, device=0, width=64, height=48)
    renderer.render_frames(16)
    accum, count = renderer.read_accumulation()
    png, hdr = str(tmp_path / "s.png"), str(tmp_path / "s.hdr")
    renderer.save_screenshot(png)
    renderer.save_screenshot(hdr)
    ldr = np.clip(accum[..., :3] * (1.0 / count), 0, 1)
    srgb = np.where(ldr <= 0.0031308, 12.92 * ldr, 1.055 * ldr ** (1 / 2.4) - 0.055)
    expected = (srgb * 255 + 0.5).astype(np.uint8)
    assert np.abs(np.asarray(Image.open(png)).astype(int) - expected).max() <= 1
    assert open(hdr, "rb").read().startswith(b"#?RADIANCE")
    renderer.close()


def test_empty_scene_renders_sky():
    scene = formats.Scene([], 0, np.ones(3, np.float32), np.zeros(3, np.float32), np.zeros((0, 2), np.uint32), np.zeros((0, 4), np.uint16), np.zeros(0, np.uint8), [])
    g = gpu_scene(scene, np.zeros((0, 4), np.float32))
    spec = synthetic.look_at_spec((0, 0, 0), (1.0, 0, 0), sky=(0.25, 0.5, 1.0), sky_strength=2.0)
    constants = oracle.write_constants(spec, 16, 8, scene, np.zeros((0, 4), np.float32))
    g.create_render_targets(16, 8)
    g.set_render_settings(3, 4, 0)
    g.render_samples(to_gpu_constants(constants), 1, 3, 1, accumulate=False)
    image, count = g.read_accumulation()
    assert count == 3
    np.testing.assert_allclose(image[..., :3], np.broadcast_to([1.5, 3.0, 6.0], (8, 16, 3)), rtol=1e-6)


def test_error_behaviour(gpu_cornell, cornell, cornell_lights, cornell_spec, capfd):
    """Nonzero return codes and a message instead of exceptions or exits (the reference's create_*() convention)."""
    lib = hip_api.lib()
    constants = to_gpu_constants(oracle.write_constants(cornell_spec, 64, 64, cornell, cornell_lights))
    gpu_cornell.create_render_targets(32, 32)
    assert lib.pt_render_samples(gpu_cornell.handle, C.byref(constants), 0, 1, 1, 0) != 0  # viewport mismatch
    assert lib.pt_set_render_settings(gpu_cornell.handle, 9, 4, 0) != 0
    assert lib.pt_set_render_settings(gpu_cornell.handle, 3, 99, 0) != 0
    assert lib.pt_create_render_targets(gpu_cornell.handle, 0, 10) != 0
    handle = C.c_void_p()
    header = hip_api.SceneHeader(1, 0)
    bad = hip_api.TextureDesc(4, 4, 37, 1, None, None)  # VK_FORMAT_R8G8B8A8_UNORM is not a supported block format
    descs = (hip_api.TextureDesc * 3)(bad, bad, bad)
    assert lib.pt_create_scene(C.byref(handle), 0, C.byref(header), None, None, None, descs, 3) != 0
    assert not handle.value
    assert lib.pt_create_scene(C.byref(handle), 0, C.byref(header), None, None, None, descs, 2) != 0  # not 3 per material
    lib.pt_destroy_scene(None)
    C.CDLL(None).fflush(None)
    out = capfd.readouterr().out
    assert "does not match" in out and "not supported" in out


def test_slideshow_cornell_path_lengths(tmp_path, cornell, cornell_lights, cornell_spec, oracle_cornell):
    """Slides 24..33 of the reference (Cornell box, path lengths 0..9) as a batch job; screenshots equal the oracle's."""
    from PIL import Image
    show = host_api.Slideshow()
    host_api.lib().create_slides(C.byref(show))
    show.slide_begin, show.slide_end = 24, 28
    frames = 3
    assert host_api.lib().run_slideshow(C.byref(show), 0, os.fsencode(DATA_ROOT), os.fsencode(str(tmp_path)), 96, 64, 1, frames) == 0
    constants = oracle.write_constants(cornell_spec, 96, 64, cornell, cornell_lights)
    for path_length in range(4):
        expected, _ = oracle_cornell.render(constants, cornell_lights, path_length=path_length, first_frame_index=2696, frame_count=frames, threads=8)
        ldr = np.clip(expected[..., :3] * np.float32(1.0 / frames), 0, 1)
        srgb = np.where(ldr <= 0.0031308, 12.92 * ldr, 1.055 * ldr ** (1 / 2.4) - 0.055)
        got = np.asarray(Image.open(str(tmp_path / "renders" / "cornell_box_path_lengths" / ("%d.png" % path_length))))
        assert got.shape == (64, 96, 3) and np.abs(got.astype(int) - (srgb * 255 + 0.5).astype(np.uint8)).max() <= 1
    assert not os.path.exists(str(tmp_path / "renders" / "cornell_box_path_lengths" / "4.png"))
    host_api.lib().free_slideshow(C.byref(show))


def test_display_image_and_jpg(tmp_path):
    from PIL import Image
    spec = host_api.quickload(os.path.join(DATA, "saves", "cornell_box", "default.rt_save"))
    renderer = host_api.Renderer(spec, DATA_ROOT, device=0, width=64, height=48)
    renderer.render_frames(8)
    accum, count = renderer.read_accumulation()
    for tonemapper in (host_api.TONEMAPPER_ACES, host_api.TONEMAPPER_KHRONOS_PBR_NEUTRAL):
        renderer.spec.tonemapper = tonemapper
        path = str(tmp_path / ("display_%d.png" % tonemapper))
        renderer.save_display_image(path)
        expected = np.zeros((48, 64, 3), np.uint8)
        host_api.lib().tonemap_image(expected.ctypes.data_as(C.c_void_p), accum.ctypes.data_as(C.c_void_p), 48 * 64, renderer.spec.exposure, count, tonemapper)
        saved = np.asarray(Image.open(path))
        assert (saved == expected).all() and expected.max() > 100
        # Independently of the host's own tonemapper: the shader's formulas in float64 followed by the sRGB encoding
        from tests.test_host_cpu import _tonemap_expected
        linear = np.array([_tonemap_expected(pixel, renderer.spec.exposure, count, tonemapper) for pixel in accum.reshape(-1, 4)]).reshape(48, 64, 3)
        linear = np.clip(linear, 0.0, 1.0)
        reference = np.where(linear <= 0.0031308, 12.92 * linear, 1.055 * linear ** (1 / 2.4) - 0.055) * 255.0
        assert np.abs(saved.astype(np.float64) - reference).max() <= 0.5 + 1e-3
    jpg = str(tmp_path / "s.jpg")
    renderer.save_screenshot(jpg)
    ldr = np.clip(accum[..., :3] * (1.0 / count), 0, 1)
    srgb = (np.where(ldr <= 0.0031308, 12.92 * ldr, 1.055 * ldr ** (1 / 2.4) - 0.055) * 255 + 0.5).astype(np.uint8)
    # a noisy image: as close to the input as another baseline encoder at the same quality gets
    other = str(tmp_path / "pil.jpg")
    Image.fromarray(srgb).save(other, quality=90, subsampling=0)
    error = np.sqrt(((np.asarray(Image.open(jpg)).astype(float) - srgb) ** 2).mean())
    assert error < 1.25 * np.sqrt(((np.asarray(Image.open(other)).astype(float) - srgb) ** 2).mean()) + 0.5
    renderer.close()


def test_full_size_results_do_not_depend_on_batching(monkeypatch):
    """1920x1080, 24 spp: one call with all frames in one wavefront batch, two calls (5 + 19 frames), and batches of two
    frames give the same bits, because every pixel adds its frames in frame order whatever the batch layout."""
    spec = host_api.quickload(os.path.join(DATA, "saves", "cornell_box", "default.rt_save"))
    images, rays = [], []
    for batch_paths, calls in ((None, (24,)), (None, (5, 19)), ("3000000", (24,))):
        if batch_paths is None:
            monkeypatch.delenv("PT_BATCH_PATHS", raising=False)
        else:
            monkeypatch.setenv("PT_BATCH_PATHS", batch_paths)
        renderer = host_api.Renderer(spec, DATA_ROOT, device=0, width=1920, height=1080)
        for count in calls:
            renderer.render_frames(count)
        image, samples = renderer.read_accumulation()
        stats = renderer.statistics()
        assert samples == 24 and renderer.spec.frame_index == spec.frame_index + 24
        images.append(image)
        rays.append(stats.extension_rays + stats.shadow_rays)
        renderer.close()
    assert (bits(images[0]) == bits(images[1])).all() and (bits(images[0]) == bits(images[2])).all()
    assert rays[0] == rays[1] == rays[2] and rays[0] > 2.5 * 24 * 1920 * 1080
    mean = images[0][..., :3].mean() / 24
    assert np.isfinite(images[0]).all() and 0.2 < mean < 2.0 and (images[0][..., 3] == 1.0).all()


def test_convergence_against_higher_sample_count_oracle(gpu_cornell, oracle_cornell, cornell, cornell_lights, cornell_spec):
    """Relative RMSE of the mean radiance against a 16 times higher-spp oracle image made of other frames: it falls like
    1 / sqrt(spp) (Monte Carlo, independent frames), so 64 spp is about twice as close as 16 spp."""
    width = height = 64
    constants = oracle.write_constants(cornell_spec, width, height, cornell, cornell_lights)
    reference, _ = oracle_cornell.render(constants, cornell_lights, first_frame_index=100000, frame_count=1024, threads=16)
    reference = reference[..., :3] / 1024.0
    gpu_cornell.create_render_targets(width, height)
    gpu_cornell.set_render_settings(hip_api.SAMPLING_STRATEGY_NEE, 4, cornell.emission_material_index)
    errors = {}
    for spp in (16, 64):
        gpu_cornell.render_samples(to_gpu_constants(constants), 2696, spp, accumulate=False)
        image, count = gpu_cornell.read_accumulation()
        mean = image[..., :3] / count
        errors[spp] = float(np.sqrt(((mean - reference) ** 2).mean()) / reference.mean())
    assert errors[64] < 0.3, errors  # a small light: 0.21 measured, the oracle image itself has about 0.05
    assert 1.5 < errors[16] / errors[64] < 2.6, errors


def test_command_line_driver(tmp_path):
    """pt_render, the headless counterpart of the reference's executable: the same pixels as the library calls it makes,
    a usage message for unknown options, and the light tree switch."""
    import subprocess
    from PIL import Image
    binary = os.path.join(os.path.dirname(host_api.LIB_PATH), "pt_render")
    assert os.path.exists(binary), "build the host first (make -C path_tracer_amd/host)"
    png = str(tmp_path / "cli.png")
    common = [binary, "-data", DATA_ROOT, "-w64", "-h48", "-spp", "16"]
    done = subprocess.run(common + ["-o", png], capture_output=True, text=True, timeout=120)
    assert done.returncode == 0, done.stdout + done.stderr
    assert "Mrays/s" in done.stdout and "Saved" in done.stdout
    spec = host_api.quickload(os.path.join(DATA, "saves", "cornell_box", "default.rt_save"))
    renderer = host_api.Renderer(spec, DATA_ROOT, device=0, width=64, height=48)
    renderer.render_frames(16)
    expected = str(tmp_path / "api.png")
    renderer.save_screenshot(expected)
    assert (np.asarray(Image.open(png)) == np.asarray(Image.open(expected))).all()
    # One light: the tree samples it with probability one, the same image up to the order of operations
    tree = str(tmp_path / "tree.png")
    done = subprocess.run(common + ["-light_tree", "-o", tree], capture_output=True, text=True, timeout=120)
    assert done.returncode == 0, done.stdout + done.stderr
    assert np.abs(np.asarray(Image.open(tree)).astype(int) - np.asarray(Image.open(png)).astype(int)).max() <= 1
    done = subprocess.run([binary, "-no_such_option"], capture_output=True, text=True, timeout=60)
    assert done.returncode != 0 and "Usage" in done.stdout
    renderer.close()


def test_committed_vectors(cornell, cornell_lights, cornell_spec):
    """The library against the committed renders of tests/golden/vectors (written by tests/golden/make_vectors.py from
    the oracle): no oracle code runs here."""
    import importlib.util
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_vectors", os.path.join(here, "make_vectors.py"))
    mv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mv)
    expected_rays = dict(line.split() for line in open(os.path.join(here, "vectors", "ray_counts.txt")))
    g_scene = gpu_scene(cornell, cornell_lights)
    constants = to_gpu_constants(oracle.write_constants(cornell_spec, mv.WIDTH, mv.HEIGHT, cornell, cornell_lights))
    g_scene.create_render_targets(mv.WIDTH, mv.HEIGHT)
    for name, strategy, light_sampling, extra in mv.CASES:
        lights = np.concatenate([cornell_lights, np.array(mv.EXTRA_LIGHTS[:extra], np.float32).reshape(-1, 4)])
        g_scene.set_lights(lights)
        g_scene.set_light_sampling(light_sampling)
        g_scene.set_render_settings(strategy, mv.PATH_LENGTH, cornell.emission_material_index)
        g_scene.reset_statistics()
        g_scene.render_samples(constants, mv.FIRST_FRAME, mv.FRAMES, 1, accumulate=False)
        got, count = g_scene.read_accumulation()
        stats = g_scene.statistics()
        expected = np.load(os.path.join(here, "vectors", "cornell_%s.npy" % name))
        assert count == mv.FRAMES and stats.extension_rays + stats.shadow_rays == int(expected_rays[name]), name
        assert (bits(got) == bits(expected)).all(), name
    g_scene.close()
