"""Pins by outputs of the reference itself (tests/golden/ref/, written by tests/golden/make_ref_fixtures.py).

Three parts of the reference compile with plain gcc in the authoring container: src/math_utilities.c, the texture
converter tools/texture_conversion/main.c (with stb_dxt.h) and src/stb_image_write.c. What they returned on committed
inputs is stored as fixtures; here the host library, the oracle, the loaders, the block decoders and the authoring
tools are compared with those results. Where oracle/_ref/ exists (the authoring container, or a GPU box that
received the built files) the math functions are also compared live on fresh inputs.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from oracle import formats
from path_tracer_amd import host_api
from path_tracer_amd.tools import vkt
from tests.test_oracle_pins import srgb_to_linear
from tests.test_texture_lod_cpu import decode_level, texture_only_scene

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "golden", "ref")
REF_BUILD = os.path.join(os.path.dirname(HERE), "oracle", "_ref")
IMAGES = ("blobs", "stripes", "noise", "flat", "pixel")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def math_vectors():
    return np.load(os.path.join(REF, "math_utilities.npz"))


def host_math():
    lib = host_api.lib()
    lib.mat_mat_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.mat_mat_mul.restype = None
    lib.normalize.argtypes = [C.c_void_p, C.c_uint32]
    lib.normalize.restype = C.c_bool
    return lib


# ---- src/math_utilities.c ----

def check_math(lib, v, invert):
    """lib: anything with the reference's four functions; invert: callable matrix -> inverse."""
    for index in range(4):
        lhs, rhs, expected = v["mul%d_lhs" % index], v["mul%d_rhs" % index], v["mul%d_out" % index]
        m, k, n = lhs.shape[1], lhs.shape[2], rhs.shape[2]
        for i in range(len(lhs)):
            out = np.zeros((m, n), np.float32)
            lib.mat_mat_mul(ptr(out), ptr(np.ascontiguousarray(lhs[i])), ptr(np.ascontiguousarray(rhs[i])), m, k, n)
            assert (bits(out) == bits(expected[i])).all()
    for i in range(len(v["normalize_in"])):
        vector = v["normalize_in"][i].copy()
        assert bool(lib.normalize(ptr(vector), 3)) == bool(v["normalize_ok"][i])
        assert (bits(vector) == bits(v["normalize_out"][i])).all()
    for i in range(len(v["angles"])):
        out = np.zeros(9, np.float32)
        lib.rotation_matrix_from_angles(ptr(out), ptr(np.ascontiguousarray(v["angles"][i])))
        assert (bits(out) == bits(v["rotations"][i])).all()
    for i in range(len(v["matrices"])):
        assert (bits(invert(v["matrices"][i])) == bits(v["inverses"][i])).all(), i


def host_invert(matrix):
    source, out = np.ascontiguousarray(matrix, np.float32).copy(), np.zeros((4, 4), np.float32)
    host_api.lib().invert_mat4(ptr(out), ptr(source))
    return out


def test_host_math_equals_the_reference_bit_for_bit(math_vectors):
    """mat_mat_mul, normalize, rotation_matrix_from_angles (on random angles and the angles of every quicksave of the
    reference), invert_mat4 (src/math_utilities.c:16-64,197-203): the host library returns the reference's bits."""
    assert len(math_vectors["angles"]) == 64 + 16 and len(math_vectors["matrices"]) == 48
    check_math(host_math(), math_vectors, host_invert)


def test_oracle_invert_mat4_equals_the_reference_bit_for_bit(math_vectors):
    for matrix, expected in zip(math_vectors["matrices"], math_vectors["inverses"]):
        assert (bits(oracle.invert_mat4(matrix)) == bits(expected)).all()


def test_reference_inverse_is_an_inverse(math_vectors):
    """Guards the fixture itself: what the reference returned inverts its input."""
    for matrix, inverse in zip(math_vectors["matrices"][:16], math_vectors["inverses"][:16]):
        np.testing.assert_allclose(inverse.astype(np.float64) @ matrix.astype(np.float64), np.eye(4), atol=5e-3)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_BUILD, "libref_math.so")), reason="oracle/_ref is built where the reference is (make -C oracle ref)")
def test_host_math_against_the_live_reference_library():
    ref = C.CDLL(os.path.join(REF_BUILD, "libref_math.so"))
    ref.normalize.restype = C.c_bool
    lib = host_math()
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    for _ in range(2000):
        angles = rng.uniform(-10, 10, 3).astype(np.float32)
        a, b = np.zeros(9, np.float32), np.zeros(9, np.float32)
        ref.rotation_matrix_from_angles(ptr(a), ptr(angles))
        lib.rotation_matrix_from_angles(ptr(b), ptr(angles))
        assert (bits(a) == bits(b)).all()
        matrix = (rng.normal(size=(4, 4)) * 10.0 ** rng.uniform(-2, 2, (4, 1))).astype(np.float32)
        if rng.random() < 0.3:
            matrix[rng.integers(0, 4), rng.integers(0, 4)] = 0.0
        expected = np.zeros((4, 4), np.float32)
        source = matrix.copy()
        ref.invert_mat4(ptr(expected), ptr(source))
        assert (bits(host_invert(matrix)) == bits(expected)).all()
        assert (bits(oracle.invert_mat4(matrix)) == bits(expected)).all()


# ---- tools/texture_conversion/main.c: files of the reference's converter ----

def source_image(name):
    from PIL import Image
    return np.asarray(Image.open(os.path.join(REF, "images", name + ".png")).convert("RGB"))


def uncompressed_levels(name, vk_format):
    """The mip levels of an RGBA8 file of the reference's converter (formats 37 and 43) as uint8 (H, W, 4)."""
    texture = formats.load_vkt(os.path.join(REF, "vkt", "%s_%d.vkt" % (name, vk_format)))
    levels = []
    for (w, h, size), offset in zip(texture.mip_extents, texture.mip_offsets):
        assert size == w * h * 4
        levels.append(texture.payload[int(offset):int(offset) + size].reshape(h, w, 4))
    return levels


@pytest.mark.parametrize("name", IMAGES)
@pytest.mark.parametrize("vk_format", [131, 132, 141])
def test_loaders_read_the_reference_converters_files(name, vk_format):
    """The numpy parser and the C loader (src/textures.c:63-106 restated) agree on the files of the reference's
    converter: extents, level count down to 4x4 (main.c:291-293), sizes, payload bytes."""
    path = os.path.join(REF, "vkt", "%s_%d.vkt" % (name, vk_format))
    texture = formats.load_vkt(path)
    width, height = source_image(name).shape[1::-1]
    if name == "pixel":
        width = height = 4  # main.c:295-307
    assert (texture.width, texture.height, texture.format) == (width, height, vk_format)
    assert texture.mip_count == min(int(np.log2(width)), int(np.log2(height))) - 1
    block_bytes = 16 if vk_format == 141 else 8
    assert [e for e in texture.mip_extents] == [(width >> m, height >> m, (width >> m) * (height >> m) // 16 * block_bytes) for m in range(texture.mip_count)]
    textures = host_api.HostTextures()
    paths = (C.c_char_p * 1)(os.fsencode(path))
    assert host_api.lib().load_textures(C.byref(textures), paths, 1) == 0
    t = textures.textures[0]
    assert (t.width, t.height, t.format, t.mipmap_count) == (width, height, vk_format, texture.mip_count)
    for m in range(texture.mip_count):
        size = texture.mip_extents[m][2]
        assert t.mipmap_sizes[m] == size
        assert C.string_at(t.mipmap_data[m], size) == texture.payload[int(texture.mip_offsets[m]):int(texture.mip_offsets[m]) + size].tobytes()
    host_api.lib().free_textures(C.byref(textures))


def oracle_levels(texture):
    """Every level of a block-compressed texture as the oracle's sampler returns it at texel centres."""
    scene = oracle.OracleScene(texture_only_scene([texture] * 3))
    levels = []
    for m, (w, h, _) in enumerate(texture.mip_extents):
        ys, xs = np.mgrid[0:h, 0:w]
        uvs = np.stack([(xs + 0.5) / w, (ys + 0.5) / h], -1).reshape(-1, 2)
        levels.append(scene.sample_texture(0, uvs, np.full(len(uvs), float(m), np.float32)).reshape(h, w, 3).astype(np.float64))
    return levels


def to_display(level, vk_format):
    """Decoded values on the 8-bit scale the converter quantised them on (sRGB-encoded for format 132)."""
    if vk_format == 132:
        level = np.where(level <= 0.0031308, 12.92 * level, 1.055 * np.power(np.maximum(level, 1e-12), 1 / 2.4) - 0.055)
    return level * 255.0


@pytest.mark.parametrize("name", ["flat", "pixel"])
def test_single_colour_blocks_of_stb_dxt_decode_to_the_colour(name):
    """stb_dxt encodes a block of one colour with endpoint pairs chosen so that the interpolated palette entry
    (2 c0 + c1) / 3 of the expanded 5:6:5 endpoints is closest to the colour (its stb__OMatch5/6 tables): a decoder
    that expands endpoints or interpolates differently misses the colour by up to 7/255. BC5 stores it exactly."""
    rgb = source_image(name)[0, 0].astype(np.float64)
    for vk_format, channels, tolerance in ((131, 3, 1.0), (132, 3, 1.0), (141, 2, 0.0)):
        texture = formats.load_vkt(os.path.join(REF, "vkt", "%s_%d.vkt" % (name, vk_format)))
        for level in oracle_levels(texture):
            decoded = to_display(level, vk_format)[..., :channels]
            assert np.abs(decoded - rgb[:channels]).max() <= tolerance + 1e-3, (vk_format, decoded[0, 0], rgb)


@pytest.mark.parametrize("name", ["blobs", "stripes", "noise"])
def test_bc5_of_stb_dxt_decodes_within_the_step_of_its_palette(name):
    """stb_compress_bc5_block spans a block's range with eight values; every texel decodes to within half a step
    (range / 14) plus rounding of what the converter quantised: the reference's own level (format 37 file)."""
    texture = formats.load_vkt(os.path.join(REF, "vkt", "%s_141.vkt" % name))
    reference_levels = uncompressed_levels(name, 37)
    for decoded, expected in zip(oracle_levels(texture), reference_levels):
        decoded, expected = decoded[..., :2] * 255.0, expected[..., :2].astype(np.float64)
        h, w, _ = expected.shape
        blocks = expected.reshape(h // 4, 4, w // 4, 4, 2)
        span = np.broadcast_to((blocks.max(axis=(1, 3)) - blocks.min(axis=(1, 3)))[:, None, :, None, :], blocks.shape).reshape(h, w, 2)
        assert (np.abs(decoded - expected) <= span / 14.0 + 1.0 + 1e-3).all()


def decode_bc1_variant(texture, level, swap_codes=False, swap_endpoints=False, replicate=True, bgr=False):
    """A BC1 decoder with switches for the mistakes a decoder can make: meaning of codes 2 and 3 exchanged, endpoints
    exchanged, 5:6:5 endpoints widened without replicating their top bits, blue in the top bits."""
    w, h, size = texture.mip_extents[level]
    payload = texture.payload[int(texture.mip_offsets[level]):int(texture.mip_offsets[level]) + size].reshape(h // 4, w // 4, 8).astype(np.int64)
    c = [payload[..., 0] | payload[..., 1] << 8, payload[..., 2] | payload[..., 3] << 8]
    codes = payload[..., 4] | payload[..., 5] << 8 | payload[..., 6] << 16 | payload[..., 7] << 24

    def expand(v):
        r, g, b = (v >> 11) & 31, (v >> 5) & 63, v & 31
        rgb = np.stack([r << 3 | r >> 2, g << 2 | g >> 4, b << 3 | b >> 2] if replicate else [r << 3, g << 2, b << 3], -1)
        return rgb[..., ::-1] if bgr else rgb
    e0, e1 = expand(c[0]), expand(c[1])
    four = (c[0] > c[1])[..., None]
    palette = [e0, e1, np.where(four, (2 * e0 + e1) // 3, (e0 + e1) // 2), np.where(four, (e0 + 2 * e1) // 3, 0)]
    if swap_codes:
        palette[2], palette[3] = palette[3], palette[2]
    if swap_endpoints:
        palette[0], palette[1] = palette[1], palette[0]
    palette = np.stack(palette, 2)  # (by, bx, 4, 3)
    image = np.zeros((h, w, 3))
    for py in range(4):
        for px in range(4):
            code = (codes >> (2 * (4 * py + px))) & 3
            image[py::4, px::4] = np.take_along_axis(palette, code[..., None, None].repeat(3, -1), 2)[:, :, 0]
    return image


@pytest.mark.parametrize("name,limit", [("blobs", 7.0), ("stripes", 10.0)])
@pytest.mark.parametrize("vk_format", [131, 132])
def test_bc1_of_stb_dxt_decodes_close_to_the_reference_level(name, vk_format, limit):
    """BC1 blocks of the reference's converter (stb_dxt, high quality mode), decoded by the oracle, against the same
    level stored uncompressed by the same converter (format 37 for 131, 43 for 132): the full-resolution level comes
    back within a few 8-bit steps (root mean square; four colours on a line per block), and the decoding rule that
    the oracle uses explains the encoder's blocks better than any of the rules a decoder could confuse it with."""
    texture = formats.load_vkt(os.path.join(REF, "vkt", "%s_%d.vkt" % (name, vk_format)))
    expected = uncompressed_levels(name, 37 if vk_format == 131 else 43)[0][..., :3].astype(np.float64)
    levels = oracle_levels(texture)
    rms = lambda image: np.sqrt(((image - expected) ** 2).mean())
    mine = rms(to_display(levels[0], vk_format))
    assert mine <= limit
    assert abs(rms(decode_bc1_variant(texture, 0)) - mine) < 1e-3
    for mistake in ({"swap_codes": True}, {"swap_endpoints": True}, {"replicate": False}, {"bgr": True}):
        assert rms(decode_bc1_variant(texture, 0, **mistake)) > mine * 1.05, mistake
    # and the independent numpy decoder returns the oracle's values on every level
    for m, level in enumerate(levels):
        np.testing.assert_allclose(level, decode_level(texture, m), atol=1e-6)


@pytest.mark.parametrize("name", ["blobs", "stripes", "noise", "flat"])
@pytest.mark.parametrize("vk_format", [37, 43])
def test_authoring_tool_mips_equal_the_reference_converters(name, vk_format):
    """path_tracer_amd/tools/vkt.py filters its mip levels like main.c:349-402 (Gaussian of standard deviation
    0.4 * 2^level over the full-resolution image, wrap-around). Level by level against the reference's uncompressed
    files: equal after 8-bit quantisation but for values that fall on a rounding boundary (at most one step)."""
    image = source_image(name).astype(np.float64)
    linear = srgb_to_linear(image) if vk_format == 43 else image / 255.0
    quantise = vkt.linear_to_srgb8 if vk_format == 43 else vkt.quantize_linear8
    levels = uncompressed_levels(name, vk_format)
    assert len(levels) == min(int(np.log2(image.shape[0])), int(np.log2(image.shape[1]))) + 1
    exact = total = 0
    for m, expected in enumerate(levels):
        assert (expected[..., 3] == 255).all()
        mine = quantise(vkt.gaussian_mipmap(linear, m)).astype(int)
        difference = np.abs(mine - expected[..., :3].astype(int))
        assert difference.max() <= 1, (name, m)
        exact += int((difference == 0).sum())
        total += difference.size
    assert exact >= 0.99 * total


@pytest.mark.parametrize("vk_format", [131, 132, 141])
def test_authoring_tool_files_match_the_reference_converters_layout(vk_format, tmp_path):
    """Same image through vkt.write_vkt() and through the reference's converter: identical headers and level
    tables; the payloads come from different block encoders and must decode close to each other."""
    image = source_image("blobs").astype(np.float64)
    linear = srgb_to_linear(image) if vk_format == 132 else image / 255.0
    path = str(tmp_path / "mine.vkt")
    vkt.write_vkt(path, linear[..., :2] if vk_format == 141 else linear, vk_format)
    mine, theirs = formats.load_vkt(path), formats.load_vkt(os.path.join(REF, "vkt", "blobs_%d.vkt" % vk_format))
    header = 32 + 24 * theirs.mip_count
    assert open(path, "rb").read()[:header] == open(os.path.join(REF, "vkt", "blobs_%d.vkt" % vk_format), "rb").read()[:header]
    assert len(mine.payload) == len(theirs.payload)
    # Both encoders against the level they compressed (the reference's uncompressed file): the range fit of vkt.py
    # loses more than stb_dxt's refined endpoints on BC1 (about 1.4 to 1.6 times the error), the same on BC5
    channels = 2 if vk_format == 141 else 3
    levels = uncompressed_levels("blobs", 43 if vk_format == 132 else 37)
    for a, b, expected in zip(oracle_levels(mine), oracle_levels(theirs), levels):
        expected = expected[..., :channels].astype(np.float64)
        error_mine = np.sqrt(((to_display(a, vk_format)[..., :channels] - expected) ** 2).mean())
        error_theirs = np.sqrt(((to_display(b, vk_format)[..., :channels] - expected) ** 2).mean())
        assert error_mine <= 1.7 * error_theirs + 1.0


# ---- src/stb_image_write.h as src/main.c:2053,2076-2077 calls it ----

def decode_rgbe_file(path):
    """Radiance .hdr -> uint8 (H, W, 4) RGBE, flat or run-length coded scanlines."""
    raw = open(path, "rb").read()
    assert raw.startswith(b"#?RADIANCE")
    end = raw.index(b"\n\n") + 2
    line_end = raw.index(b"\n", end)
    tokens = raw[end:line_end].split()
    assert tokens[0] == b"-Y" and tokens[2] == b"+X"
    height, width = int(tokens[1]), int(tokens[3])
    data = raw[line_end + 1:]
    out = np.zeros((height, width, 4), np.uint8)
    position = 0
    for y in range(height):
        if 8 <= width < 32768 and data[position] == 2 and data[position + 1] == 2 and (data[position + 2] << 8 | data[position + 3]) == width:
            position += 4
            for channel in range(4):
                x = 0
                while x < width:
                    count = data[position]
                    position += 1
                    if count > 128:
                        out[y, x:x + count - 128, channel] = data[position]
                        position += 1
                        x += count - 128
                    else:
                        out[y, x:x + count, channel] = np.frombuffer(data[position:position + count], np.uint8)
                        position += count
                        x += count
        else:
            out[y] = np.frombuffer(data[position:position + 4 * width], np.uint8).reshape(width, 4)
            position += 4 * width
    assert position == len(data)
    return out


def test_hdr_screenshot_pixels_equal_stb_image_write(tmp_path):
    """write_hdr() stores the RGBE quadruples that stbi_write_hdr() stores (the reference's file is run-length coded,
    this one is flat: both are valid Radiance files that decode to the same bytes)."""
    hdr = np.load(os.path.join(REF, "stb", "hdr_pixels.npy"))
    path = str(tmp_path / "mine.hdr")
    assert host_api.lib().write_hdr(os.fsencode(path), hdr.shape[1], hdr.shape[0], ptr(hdr)) == 0
    mine, theirs = decode_rgbe_file(path), decode_rgbe_file(os.path.join(REF, "stb", "image.hdr"))
    assert (mine == theirs).all()


def test_png_screenshot_pixels_equal_stb_image_write(tmp_path):
    from PIL import Image
    ldr = np.load(os.path.join(REF, "stb", "ldr_pixels.npy"))
    path = str(tmp_path / "mine.png")
    assert host_api.lib().write_png(os.fsencode(path), ldr.shape[1], ldr.shape[0], ptr(ldr)) == 0
    theirs = np.asarray(Image.open(os.path.join(REF, "stb", "image.png")))
    assert (theirs == ldr).all() and (np.asarray(Image.open(path)) == theirs).all()


def jpeg_segments(path):
    raw = open(path, "rb").read()
    assert raw[:2] == b"\xff\xd8"
    position, segments = 2, {}
    while raw[position + 1] != 0xda:
        marker, size = raw[position + 1], int.from_bytes(raw[position + 2:position + 4], "big")
        segments.setdefault(marker, []).append(raw[position + 4:position + 2 + size])
        position += 2 + size
    return segments


def test_jpg_screenshot_against_stb_image_write(tmp_path):
    """Quality 90 as src/main.c:2076 asks for it: same quantisation tables as stb_image_write's file, hence the same
    quality; the decoded images agree to within the rounding of two different forward transforms."""
    from PIL import Image
    ldr = np.load(os.path.join(REF, "stb", "ldr_pixels.npy"))
    path = str(tmp_path / "mine.jpg")
    assert host_api.lib().write_jpg(os.fsencode(path), ldr.shape[1], ldr.shape[0], ptr(ldr), 90) == 0
    mine, theirs = jpeg_segments(path), jpeg_segments(os.path.join(REF, "stb", "image.jpg"))
    assert b"".join(mine[0xdb]) == b"".join(theirs[0xdb])  # DQT
    a = np.asarray(Image.open(path)).astype(int)
    b = np.asarray(Image.open(os.path.join(REF, "stb", "image.jpg"))).astype(int)
    assert a.shape == b.shape == ldr.shape
    assert np.sqrt(((a - b) ** 2).mean()) < 2.0
    assert abs(np.sqrt(((a - ldr) ** 2).mean()) - np.sqrt(((b - ldr) ** 2).mean())) < 1.0


# ---- the reference converter's textures through the shading path (CPU: oracle; GPU: tests/test_gpu_reference_pins.py) ----

def reference_textured_quad(name="blobs"):
    """A unit-uv square whose material carries the reference converter's BC1 sRGB base colour, BC1 specular and BC5
    normal files of one image; returns the scene, hit records at the texel centres of level 0 and the reference's
    own uncompressed levels 0 (sRGB file and UNORM file) to compare with."""
    from tests.test_texture_lod_cpu import textured_quad
    base = formats.load_vkt(os.path.join(REF, "vkt", "%s_132.vkt" % name))
    scene, _ = textured_quad(base.width, 1.0, 2.0)
    scene.textures[3] = base
    scene.textures[4] = formats.load_vkt(os.path.join(REF, "vkt", "%s_131.vkt" % name))
    size = base.width
    ys, xs = np.mgrid[0:size, 0:size]
    u, v = ((xs + 0.5) / size).reshape(-1), ((ys + 0.5) / size).reshape(-1)
    # The square is two triangles (corners 0, 1, 2 and 0, 2, 3 with uv (0,0), (1,0), (1,1), (0,1)); barycentrics of
    # the second and third vertex
    lower = u >= v
    tri = np.where(lower, 0, 1).astype(np.int32)
    barys = np.where(lower[:, None], np.stack([u - v, v], 1), np.stack([u, v - u], 1)).astype(np.float32)
    return scene, tri, barys, uncompressed_levels(name, 43)[0], uncompressed_levels(name, 37)[0]


def check_reference_textures_through_shading(shading_data, scene, tri, barys, srgb_level, unorm_level):
    """shading_data(tri, barys, front, out_dirs) -> (n, 20). Roughness is the squared green channel of the specular
    texture; with the metalness (its blue channel) known, the diffuse albedo gives the base colour back
    (src/shaders/shading_data.glsl:96-100)."""
    n = len(tri)
    out = shading_data(tri, barys, np.ones(n, np.uint8), np.tile([0.0, 0.0, 1.0], (n, 1)).astype(np.float32)).astype(np.float64)
    size = srgb_level.shape[0]
    texture = scene.textures[4]
    specular = decode_level(texture, 0).reshape(-1, 3)
    roughness = np.maximum(0.006, specular[:, 1] ** 2)
    # 16-bit texture coordinates put a lookup up to 1/100 of a texel beside the centre: a small blend with neighbours
    assert np.abs(out[:, 19] - roughness).max() < 0.02 and np.sqrt(((out[:, 19] - roughness) ** 2).mean()) < 2e-3
    metal = specular[:, 2]
    keep = metal < 0.9
    base = out[keep, 13:16] / (1.0 - metal[keep, None])
    expected = srgb_to_linear(srgb_level[..., :3].astype(np.float64)).reshape(-1, 3)[keep]
    rms = np.sqrt(((to_display(np.clip(base, 0, 1), 132) - to_display(expected, 132)) ** 2).mean())
    assert rms <= 7.5, rms
    # and the specular texture against the reference's uncompressed level
    green = unorm_level[..., 1].reshape(-1).astype(np.float64)
    bright = green >= 40.0  # below, the clamp of the roughness at 0.006 hides the texture
    rms = np.sqrt(((np.sqrt(out[bright, 19]) * 255.0 - green[bright]) ** 2).mean())
    assert rms <= 7.5, rms
    return out


def test_reference_converter_textures_through_the_oracles_shading_path():
    scene, tri, barys, srgb_level, unorm_level = reference_textured_quad()
    o_scene = oracle.OracleScene(scene)
    constants = oracle.write_constants(formats.default_scene_spec(), 32, 32, scene, np.zeros((0, 4), np.float32))
    check_reference_textures_through_shading(lambda t, b, f, o: o_scene.shading_data(constants, t, b, f, o), scene, tri, barys, srgb_level, unorm_level)
