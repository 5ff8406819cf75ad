"""CPU-side checks of the product's host code: the C ABI library exports what its header declares, the C host
agrees with the oracle's independent restatements (bitwise for matrices and constants) and with the numpy file
parsers. Nothing here launches GPU work."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from oracle import formats
from path_tracer_amd import hip_api, host_api, multi_gpu
from tests.paths import DATA, DATA_ROOT, ROOT


def declared_functions(header_path):
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|void\*?|void|uint32_t|double|pt_scene_t\*)\s+\*?\s*(pt_[a-z_0-9]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    names = declared_functions(os.path.join(ROOT, "include", "pt_hip.h"))
    assert len(names) >= 18
    assert sorted(names) == sorted(hip_api.EXPORTED_SYMBOLS)
    library = C.CDLL(hip_api.LIB_PATH)
    for name in names:
        assert hasattr(library, name), name


def test_host_library_exports():
    library = host_api.lib()
    for name in host_api.EXPORTED_SYMBOLS:
        assert hasattr(library, name), name


def test_abi_struct_sizes():
    assert C.sizeof(hip_api.Constants) == 784
    assert C.sizeof(host_api.SceneSpec) == 128
    assert C.sizeof(hip_api.Statistics) == 8 * 8 + 5 * 8 + 5 * 8 + 2 * 4 + 2 * 8
    assert hip_api.lib().pt_abi_version() == hip_api.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "pt_hip.h")).read()
    assert int(re.search(r"#define PT_HIP_ABI_VERSION (\d+)u", header).group(1)) == hip_api.ABI_VERSION


def test_invert_mat4_bitwise_against_oracle():
    rng = np.random.default_rng(11)
    for _ in range(50):
        m = rng.normal(size=16).astype(np.float32)
        out = np.zeros(16, np.float32)
        host_api.lib().invert_mat4(out.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p))
        assert (out.view(np.uint32) == oracle.invert_mat4(m).reshape(16).view(np.uint32)).all()


def all_saves():
    saves = []
    for root, _, files in os.walk(os.path.join(DATA, "saves")):
        saves += [os.path.join(root, f) for f in sorted(files) if f.endswith(".rt_save")]
    return sorted(saves)


@pytest.mark.parametrize("save", all_saves())
def test_quickload_and_constants_bitwise(save, cornell):
    """Every quicksave of the reference: C host == numpy parser; constants block == oracle restatement, bit for bit."""
    spec = host_api.quickload(save)
    parsed = formats.load_rt_save(save)
    assert bytes(spec) == parsed.tobytes()
    lights = formats.load_lights(os.path.join(DATA, "Bistro_outside.lights"))
    lit = host_api.LitScene()
    lit.scene.header.dequantization_factor[:] = [float(v) for v in cornell.dequantization_factor]
    lit.scene.header.dequantization_summand[:] = [float(v) for v in cornell.dequantization_summand]
    light_array = ((C.c_float * 4) * len(lights))()
    for i, l in enumerate(lights):
        light_array[i][:] = [float(v) for v in l]
    lit.spherical_lights = C.cast(light_array, C.POINTER(C.c_float * 4))
    lit.spherical_light_count = len(lights)
    for width, height in ((1920, 1080), (512, 512), (50, 37)):
        targets = host_api.RenderTargets(width, height, 3)
        constants = hip_api.Constants()
        assert host_api.lib().write_constant_buffer(C.byref(constants), C.byref(spec), C.byref(lit), C.byref(targets)) == 0
        expected = oracle.write_constants(parsed, width, height, cornell, lights, accum_frame_count=3)
        assert bytes(constants) == bytes(expected)


def test_default_scene_spec_matches_parser():
    spec = host_api.SceneSpec()
    host_api.lib().init_scene_spec(C.byref(spec))
    assert bytes(spec) == formats.default_scene_spec().tobytes()
    settings = host_api.RenderSettings()
    host_api.lib().init_render_settings(C.byref(settings))
    assert (settings.sampling_strategy, settings.path_length) == (3, 4)


def test_scene_file_table():
    paths = host_api.scene_file_paths(1)
    assert paths == {"name": "Cornell box", "scene": "data/cornell_box.vks", "textures": "data/cornell_box_textures",
                     "lights": "data/cornell_box.lights", "quicksave": "data/saves/cornell_box/default.rt_save"}
    assert host_api.scene_file_paths(0)["lights"] == "data/Bistro_outside.lights"
    outs = [C.c_char_p() for _ in range(5)]
    assert host_api.lib().get_scene_file(99, *[C.byref(o) for o in outs]) != 0


def test_texture_loader_against_parser():
    directory = os.path.join(DATA, "cornell_box_textures")
    names = sorted(os.listdir(directory))
    paths = (C.c_char_p * len(names))(*[os.fsencode(os.path.join(directory, n)) for n in names])
    textures = host_api.HostTextures()
    assert host_api.lib().load_textures(C.byref(textures), paths, len(names)) == 0
    for i, name in enumerate(names):
        parsed = formats.load_vkt(os.path.join(directory, name))
        t = textures.textures[i]
        assert (t.width, t.height, t.format, t.mipmap_count) == (parsed.width, parsed.height, parsed.format, parsed.mip_count)
        for m in range(t.mipmap_count):
            size = t.mipmap_sizes[m]
            assert size == parsed.mip_extents[m][2]
            data = C.string_at(t.mipmap_data[m], size)
            offset = int(parsed.mip_offsets[m])
            assert data == parsed.payload[offset:offset + size].tobytes()
    host_api.lib().free_textures(C.byref(textures))
    assert textures.texture_count == 0


def test_loaders_reject_bad_files(tmp_path, capfd):
    bad = tmp_path / "bad.vkt"
    bad.write_bytes(b"\x00" * 64)
    paths = (C.c_char_p * 1)(os.fsencode(str(bad)))
    textures = host_api.HostTextures()
    assert host_api.lib().load_textures(C.byref(textures), paths, 1) != 0
    missing = (C.c_char_p * 1)(os.fsencode(str(tmp_path / "missing.vkt")))
    assert host_api.lib().load_textures(C.byref(textures), missing, 1) != 0
    scene = host_api.HostScene()
    bad_scene = tmp_path / "bad.vks"
    bad_scene.write_bytes(b"\x01\x02\x03\x04" * 8)
    assert host_api.lib().load_scene(C.byref(scene), 0, os.fsencode(str(bad_scene)), b".") != 0
    assert host_api.lib().load_scene(C.byref(scene), 0, os.fsencode(str(tmp_path / "none.vks")), b".") != 0
    assert not scene.device_scene
    spec = host_api.SceneSpec()
    assert host_api.lib().quickload(C.byref(spec), os.fsencode(str(tmp_path / "none.rt_save"))) != 0
    C.CDLL(None).fflush(None)  # the C host prints through libc's buffered stdout
    out = capfd.readouterr().out
    assert "not a valid *.vks file" in out and "Could not open the quicksave file" in out


def test_quicksave_round_trip(tmp_path):
    spec = host_api.quickload(os.path.join(DATA, "saves", "bistro", "outside.rt_save"))
    path = str(tmp_path / "copy.rt_save")
    assert host_api.lib().quicksave(C.byref(spec), os.fsencode(path)) == 0
    assert open(path, "rb").read() == open(os.path.join(DATA, "saves", "bistro", "outside.rt_save"), "rb").read()


def test_image_writers(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    path = str(tmp_path / "a.png")
    assert host_api.lib().write_png(os.fsencode(path), 53, 37, rgb.ctypes.data_as(C.c_void_p)) == 0
    assert (np.asarray(Image.open(path)) == rgb).all()
    big = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)  # noise: literals only, longer than the match window
    assert host_api.lib().write_png(os.fsencode(path), 400, 300, big.ctypes.data_as(C.c_void_p)) == 0
    assert (np.asarray(Image.open(path)) == big).all()
    # Gradients, flat areas and repeated tiles exercise filters and matches of every length and distance class
    y, x = np.mgrid[0:300, 0:400]
    smooth = np.stack([x * 255 // 399, y * 255 // 299, (x + y) * 255 // 698], axis=2).astype(np.uint8)
    tiles = np.tile(rng.integers(0, 256, (7, 5, 3), dtype=np.uint8), (60, 90, 1))
    far = np.concatenate([big[:40], smooth[:100], big[:40]])  # the second copy of the noise lies beyond the 32 KB window
    for image, largest in ((smooth, 20000), (tiles, 20000), (np.full((64, 64, 3), 77, np.uint8), 400), (rng.integers(0, 256, (1, 1, 3), dtype=np.uint8), 100),
                           (rng.integers(0, 4, (1000, 3, 3), dtype=np.uint8), 9000), (far, far.nbytes * 9 // 8)):
        image = np.ascontiguousarray(image)
        assert host_api.lib().write_png(os.fsencode(path), image.shape[1], image.shape[0], image.ctypes.data_as(C.c_void_p)) == 0
        assert (np.asarray(Image.open(path)) == image).all()
        assert os.path.getsize(path) <= largest
    hdr = (rng.random((20, 30, 3)) * np.array([0.01, 5.0, 300.0])).astype(np.float32)
    hdr[0, 0] = 0
    path = str(tmp_path / "a.hdr")
    assert host_api.lib().write_hdr(os.fsencode(path), 30, 20, hdr.ctypes.data_as(C.c_void_p)) == 0
    raw = open(path, "rb").read()
    header_end = raw.index(b"+X 30\n") + 6
    assert raw.startswith(b"#?RADIANCE\n") and b"-Y 20 +X 30" in raw
    rgbe = np.frombuffer(raw[header_end:], np.uint8).reshape(20, 30, 4).astype(np.float64)
    decoded = rgbe[..., :3] * np.exp2(rgbe[..., 3:] - 136.0)
    decoded[rgbe[..., 3] == 0] = 0
    # RGBE keeps 8 bits of the largest component
    assert np.abs(decoded - hdr).max(axis=2).max() <= np.abs(hdr).max() / 128
    assert (np.abs(decoded - hdr).max(axis=2) <= hdr.max(axis=2) / 100 + 1e-30).all()


def test_frame_slices_partition_the_sample_set():
    for world in (1, 2, 3, 4, 8):
        for total in (0, 1, 5, 8, 64, 1023):
            covered = []
            for rank in range(world):
                first, count, stride = multi_gpu.frame_slice(rank, world, 2696, total)
                assert stride == world and first == 2696 + rank
                covered += [first + i * stride for i in range(count)]
            assert sorted(covered) == list(range(2696, 2696 + total))
    with pytest.raises(ValueError):
        multi_gpu.frame_slice(4, 4, 0, 10)


def test_library_frame_slices_partition_the_sample_set():
    """What pt_context_render_samples() hands each GPU (pt_context_frame_slice): the union of the slices is the
    sample set, each frame once, also when there are fewer frames than GPUs, and it equals the Python slices that the
    one-process-per-GPU path uses."""
    lib = hip_api.lib()
    for gpus in (1, 2, 3, 4, 7, 8):
        for total in (0, 1, 2, 5, 7, 8, 9, 64, 1023):
            covered = []
            for gpu in range(gpus):
                first, count = C.c_uint32(), C.c_uint32()
                assert lib.pt_context_frame_slice(gpu, gpus, 816, total, C.byref(first), C.byref(count)) == 0
                assert (first.value, count.value, gpus) == multi_gpu.frame_slice(gpu, gpus, 816, total)
                if total <= gpu:
                    assert count.value == 0
                covered += [first.value + i * gpus for i in range(count.value)]
            assert sorted(covered) == list(range(816, 816 + total))
    out = C.c_uint32()
    assert lib.pt_context_frame_slice(2, 2, 0, 4, C.byref(out), C.byref(out)) != 0
    assert lib.pt_context_frame_slice(0, 0, 0, 4, C.byref(out), C.byref(out)) != 0
    assert lib.pt_context_frame_slice(0, 1, 0, 4, None, C.byref(out)) != 0


def test_bench_refuses_to_run_without_the_gpus_it_was_asked_for():
    """No GPU here: the benchmark must say so and fail, whatever --gpus asks for (on a box with one GPU the GPU
    suite checks that --gpus 2 is refused likewise)."""
    import subprocess
    import sys
    for gpus in ("1", "2"):
        result = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus, "--no-cpu-baseline", "--workload", "cornell"], capture_output=True, text=True, timeout=300)
        if hip_api.lib().pt_get_device_count() == 0:
            assert result.returncode != 0 and "needs a GPU" in result.stderr and result.stdout.strip() == ""


def test_write_jpg_decodes_close_to_input(tmp_path):
    from PIL import Image
    y, x = np.mgrid[0:45, 0:70]
    rgb = np.stack([(x * 3) % 256, (y * 5) % 256, (x + y) * 2 % 256], axis=2).astype(np.uint8)
    rgb[10:30, 20:50] = (200, 30, 60)
    path = str(tmp_path / "a.jpg")
    assert host_api.lib().write_jpg(os.fsencode(path), 70, 45, rgb.ctypes.data_as(C.c_void_p), 90) == 0
    image = Image.open(path)
    assert image.size == (70, 45) and image.mode == "RGB"
    decoded = np.asarray(image).astype(int)
    assert np.abs(decoded - rgb)[16:24, 24:40].max() <= 3  # blocks that lie inside the flat area
    assert np.sqrt(((decoded - rgb) ** 2).mean()) < 3.0
    low = str(tmp_path / "b.jpg")
    assert host_api.lib().write_jpg(os.fsencode(low), 70, 45, rgb.ctypes.data_as(C.c_void_p), 20) == 0
    assert os.path.getsize(low) < os.path.getsize(path)
    assert host_api.lib().write_jpg(os.fsencode(str(tmp_path / "no" / "c.jpg")), 70, 45, rgb.ctypes.data_as(C.c_void_p), 90) != 0


def _tonemap_expected(hdr, exposure, count, tonemapper):
    """tonemap.frag.glsl:17-82 in float64."""
    color = hdr[:3].astype(np.float64) * (exposure / count)
    if tonemapper == 0:
        color = np.clip(color, 0.0, 1.0)
    elif tonemapper == 1:
        aces_in = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
        aces_out = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
        v = aces_in @ color
        color = aces_out @ ((v * (v + 0.0245786) - 0.000090537) / (v * (0.983729 * v + 0.4329510) + 0.238081))
    elif tonemapper == 2:
        darkest = color.min()
        color = color - (darkest - 6.25 * darkest * darkest if darkest < 0.08 else 0.04)
        brightest = color.max()
        if brightest >= 0.76:
            compressed = 0.24
            new_brightest = 1.0 - compressed * compressed / (brightest + compressed - 0.76)
            color = color * (new_brightest / brightest)
            weight = 1.0 - 1.0 / (0.15 * (brightest - new_brightest) + 1.0)
            color = color * (1.0 - weight) + new_brightest * weight
    if np.isnan(hdr).any():
        color = np.array([1.0, 0.0, 1.0])
    if np.isinf(hdr).any():
        color = np.array([1.0, 0.0, 0.0])
    return color


def test_tonemappers_match_the_shader_formulas():
    rng = np.random.default_rng(11)
    pixels = np.concatenate([rng.uniform(0.0, 0.2, (50, 4)), rng.uniform(0.0, 30.0, (50, 4)), rng.uniform(0.0, 600.0, (20, 4))]).astype(np.float32)
    pixels[:, 3] = 7.0
    pixels = np.concatenate([pixels, np.array([[np.nan, 1, 1, 7], [1, np.inf, 1, 7], [np.nan, np.inf, 0, 7], [0, 0, 0, 7]], np.float32)])
    for tonemapper in (host_api.TONEMAPPER_CLAMP, host_api.TONEMAPPER_ACES, host_api.TONEMAPPER_KHRONOS_PBR_NEUTRAL):
        for hdr in pixels:
            out = np.zeros(3, np.float32)
            host_api.lib().tonemap_pixel(out.ctypes.data_as(C.POINTER(C.c_float)), hdr.ctypes.data_as(C.POINTER(C.c_float)), 1.5, 7, tonemapper)
            expected = _tonemap_expected(hdr, 1.5, 7, tonemapper)
            np.testing.assert_allclose(out, expected, rtol=2e-5, atol=2e-6)
        image = np.zeros((len(pixels), 3), np.uint8)
        host_api.lib().tonemap_image(image.ctypes.data_as(C.c_void_p), pixels.ctypes.data_as(C.c_void_p), len(pixels), 1.5, 7, tonemapper)
        linear = np.clip(np.array([_tonemap_expected(hdr, 1.5, 7, tonemapper) for hdr in pixels]), 0.0, 1.0)
        srgb = np.where(linear <= 0.0031308, 12.92 * linear, 1.055 * linear ** (1.0 / 2.4) - 0.055)
        assert np.abs(image.astype(int) - (srgb * 255.0 + 0.5).astype(int)).max() <= 1
    assert tuple(image[-4]) == (255, 0, 255) and tuple(image[-3]) == (255, 0, 0) and tuple(image[-2]) == (255, 0, 0)


def test_slides_table():
    """The 44 slides of src/slides.c: 14 single views and three series of ten path lengths."""
    show = host_api.Slideshow()
    assert host_api.lib().create_slides(C.byref(show)) == 44 and show.slide_count == 44
    first, hdr, cornell = show.slides[0], show.slides[13], show.slides[24 + 3]
    assert first.quicksave == b"data/saves/bistro/pretty.rt_save" and first.screenshot_frame == 2048
    assert (first.render_settings.sampling_strategy, first.render_settings.path_length) == (3, 4)
    assert first.screenshot_path == b"renders/pretty_bistro.png" and first.screenshot_format == host_api.IMAGE_FILE_FORMAT_PNG
    assert hdr.screenshot_format == host_api.IMAGE_FILE_FORMAT_HDR and hdr.screenshot_frame == 1
    assert cornell.quicksave == b"data/saves/cornell_box/default.rt_save" and cornell.render_settings.path_length == 3
    assert cornell.screenshot_path == b"renders/cornell_box_path_lengths/3.png" and cornell.screenshot_frame == 2048
    assert show.slides[5].render_settings.sampling_strategy == 0 and show.slides[6].render_settings.sampling_strategy == 1
    assert show.slides[14].screenshot_frame == 16 * 2048 and show.slides[43].quicksave == b"data/saves/living_room/day.rt_save"
    show.slide_begin, show.slide_end = 0, 45
    assert host_api.lib().run_slideshow(C.byref(show), 0, b".", b".", 64, 64, 0, 1) != 0  # past the end: refused before any GPU work
    host_api.lib().free_slideshow(C.byref(show))
    assert show.slide_count == 0


def test_scene_with_material_index_out_of_range_is_refused():
    """A triangle whose material index lies beyond the material table must make pt_create_scene() return
    nonzero before anything reaches the GPU (the check runs on the host, so it is testable without one)."""
    from tests import synthetic
    scene = synthetic.triangle_soup(seed=3, triangle_count=50, material_count=2)
    textures = [(t.width, t.height, t.format, [t.payload[int(t.mip_offsets[m]):int(t.mip_offsets[m]) + t.mip_extents[m][2]] for m in range(t.mip_count)])
                for t in scene.textures]
    materials = scene.materials.copy()
    materials[17] = 2
    with pytest.raises(hip_api.PtError):
        hip_api.Scene(2, scene.dequantization_factor, scene.dequantization_summand, scene.positions, scene.normals_uvs, materials, textures)
    # Triangles but no materials at all
    with pytest.raises(hip_api.PtError):
        hip_api.Scene(0, scene.dequantization_factor, scene.dequantization_summand, scene.positions, scene.normals_uvs, scene.materials, [])


def test_texture_file_with_wrapping_mip_header_is_refused(tmp_path):
    """offset + size of a mip header that wraps around 2^64 must not pass the bounds check of the loader, nor may a
    payload size beyond the end of the file."""
    import struct
    source = open(os.path.join(DATA, "cornell_box_textures", "white_BaseColor.vkt"), "rb").read()
    marker, version, mips, width, height, fmt, payload = struct.unpack_from("<IIIIIIQ", source, 0)
    assert marker == 0xbc1bc1 and mips >= 1
    library = host_api.lib()

    def loads(blob):
        path = tmp_path / "t.vkt"
        path.write_bytes(blob)
        textures = host_api.HostTextures()
        paths = (C.c_char_p * 1)(str(path).encode())
        code = library.load_textures(C.byref(textures), paths, 1)
        if code == 0:
            library.free_textures(C.byref(textures))
        return code == 0

    assert loads(source)
    # first mip header: u32 w, u32 h, u64 size, u64 offset at byte 32
    wrapped = bytearray(source)
    struct.pack_into("<QQ", wrapped, 32 + 8, 16, 2 ** 64 - 8)
    assert not loads(bytes(wrapped))
    huge = bytearray(source)
    struct.pack_into("<Q", huge, 24, 2 ** 40)
    assert not loads(bytes(huge))


def test_textures_with_inconsistent_mip_chains_are_refused():
    """Levels end at an extent of 4 and level m holds (width >> m) x (height >> m) texels: anything else is refused
    on the host, before the payloads are sent to the GPU."""
    from tests import synthetic
    scene = synthetic.triangle_soup(seed=3, triangle_count=10, material_count=1, texture_size=16)

    def create(textures):
        return hip_api.Scene(1, scene.dequantization_factor, scene.dequantization_summand, scene.positions, scene.normals_uvs, scene.materials, textures)

    good = [(t.width, t.height, t.format, [t.payload[int(t.mip_offsets[m]):int(t.mip_offsets[m]) + t.mip_extents[m][2]] for m in range(t.mip_count)]) for t in scene.textures]
    assert [len(t[3]) for t in good] == [3, 3, 3]
    too_many = [(w, h, f, mips + [mips[-1]]) for w, h, f, mips in good]
    with pytest.raises(hip_api.PtError):
        create(too_many)
    wrong_size = [(w, h, f, [mips[0], mips[1][:-8], mips[2]]) for w, h, f, mips in good]
    with pytest.raises(hip_api.PtError):
        create(wrong_size)
