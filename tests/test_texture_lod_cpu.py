"""Mip chains, trilinear filtering and the level-of-detail rule (ray cones), on the CPU: the oracle against an
independent numpy restatement of the filter and against the closed form of the rule, and the kernel bodies
(tests/emu) against the oracle bit for bit. The reference leaves filtering and the level of detail to the Vulkan
driver (sampler at src/main.c:864-873, texture() at src/shaders/shading_data.glsl:74-76), so these tests pin this
build's definition, not an output of the reference."""
import numpy as np
import pytest

import oracle
from oracle import formats
from tests import emu, synthetic
from tests.test_oracle_pins import bc1_texel, bc4_texel, srgb_to_linear


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def decode_level(texture, level):
    """One mip level as float64 values in [0, 1] (sRGB decoded for format 132), by the numpy block decoders."""
    w, h, size = texture.mip_extents[level]
    block_bytes = 16 if texture.format == 141 else 8
    payload = bytes(texture.payload[int(texture.mip_offsets[level]):int(texture.mip_offsets[level]) + size])
    image = np.zeros((h, w, 3))
    for y in range(h):
        for x in range(w):
            block = payload[((y // 4) * (w // 4) + x // 4) * block_bytes:][:block_bytes]
            if texture.format == 141:
                image[y, x] = [bc4_texel(block[:8], x & 3, y & 3) / 255.0, bc4_texel(block[8:], x & 3, y & 3) / 255.0, 0.0]
            else:
                rgb = bc1_texel(block, x & 3, y & 3)
                image[y, x] = srgb_to_linear(rgb) if texture.format == 132 else rgb / 255.0
    return image


def bilinear(image, uv):
    h, w = image.shape[:2]
    x, y = uv[0] * w - 0.5, uv[1] * h - 0.5
    x0, y0 = int(np.floor(x)), int(np.floor(y))
    wx, wy = x - x0, y - y0
    t = lambda xx, yy: image[yy % h, xx % w]
    return (t(x0, y0) * (1 - wx) + t(x0 + 1, y0) * wx) * (1 - wy) + (t(x0, y0 + 1) * (1 - wx) + t(x0 + 1, y0 + 1) * wx) * wy


def texture_only_scene(textures):
    return formats.Scene(["m"], 0, np.ones(3, np.float32), np.zeros(3, np.float32), np.zeros((0, 2), np.uint32), np.zeros((0, 4), np.uint16), np.zeros(0, np.uint8), textures)


@pytest.mark.parametrize("fmt", [131, 132, 141])
def test_trilinear_against_numpy(fmt):
    rng = np.random.default_rng(fmt + 7)
    texture = synthetic.random_texture(rng, 32, fmt)
    assert texture.mip_count == 4 and [e[0] for e in texture.mip_extents] == [32, 16, 8, 4]
    levels = [decode_level(texture, m) for m in range(texture.mip_count)]
    s = oracle.OracleScene(texture_only_scene([texture] * 3))
    uvs = rng.uniform(-2, 3, (300, 2)).astype(np.float32)
    lods = rng.uniform(-1, 4.5, 300).astype(np.float32)
    lods[:8] = [0, 1, 2, 3, 3.5, -0.5, 2.999, 0.001]
    got = s.sample_texture(0, uvs, lods)
    for uv, lod, value in zip(uvs.astype(np.float64), lods.astype(np.float64), got):
        lod = min(max(lod, 0.0), 3.0)
        level = int(np.floor(lod))
        weight = lod - level
        expected = bilinear(levels[level], uv)
        if weight > 0:
            expected = expected * (1 - weight) + bilinear(levels[level + 1], uv) * weight
        np.testing.assert_allclose(value, expected, atol=2e-5)


def level_coded_texture(size, fmt=132):
    """Every mip level is one colour, a different one per level: a lookup returns the level it was taken from.
    BC1 endpoints with all five red bits the same pattern decode to exact 8-bit values."""
    block_bytes = 8
    mips = int(np.log2(size)) - 1
    payloads, extents, offsets, offset, values = [], [], [], 0, []
    for m in range(mips):
        s = size >> m
        red5 = 4 + 3 * m
        values.append((red5 << 3) | (red5 >> 2))
        c = red5 << 11
        block = c.to_bytes(2, "little") + c.to_bytes(2, "little") + bytes(4)
        data = np.frombuffer(block * ((s // 4) * (s // 4)), np.uint8)
        payloads.append(data)
        extents.append((s, s, len(data)))
        offsets.append(offset)
        offset += len(data)
    return formats.Texture(size, size, fmt, mips, extents, np.array(offsets, np.uint64), np.concatenate(payloads)), values


def constant_texture(size, fmt, rgb565=0, bc4=(128, 128)):
    mips = int(np.log2(size)) - 1
    payloads, extents, offsets, offset = [], [], [], 0
    for m in range(mips):
        s = size >> m
        if fmt == 141:
            block = bytes([bc4[0], bc4[0]]) + bytes(6) + bytes([bc4[1], bc4[1]]) + bytes(6)
        else:
            block = rgb565.to_bytes(2, "little") * 2 + bytes(4)
        data = np.frombuffer(block * ((s // 4) * (s // 4)), np.uint8)
        payloads.append(data); extents.append((s, s, len(data))); offsets.append(offset); offset += len(data)
    return formats.Texture(size, size, fmt, mips, extents, np.array(offsets, np.uint64), np.concatenate(payloads))


def textured_quad(texture_size=64, uv_repeats=1.0, edge=2.0):
    """A square of edge length `edge` in the plane z = 0 (normal +z) whose texture coordinates cover uv_repeats
    of the texture per edge; base colour texture level-coded, no metal, flat normal map."""
    corners = np.array([[0, 0, 0], [edge, 0, 0], [edge, edge, 0], [0, edge, 0]], np.float64)
    tris = np.array([[corners[0], corners[1], corners[2]], [corners[0], corners[2], corners[3]]])
    uv_corners = np.array([[0, 0], [uv_repeats, 0], [uv_repeats, uv_repeats], [0, uv_repeats]], np.float64)
    tri_uvs = np.array([[uv_corners[0], uv_corners[1], uv_corners[2]], [uv_corners[0], uv_corners[2], uv_corners[3]]])
    # pad the bounding box so that the quantisation grid is not degenerate along z
    points = np.concatenate([tris.reshape(-1, 3), [[0, 0, -1], [edge, edge, 1]]])
    positions, factor, summand = synthetic.quantize_positions(points)
    positions = positions[:6]
    normals = np.tile([0.0, 0.0, 1.0], (6, 1))
    # stored uv = uv * 65535 / 8 with v flipped on decode: uv = t * (8, -8) + (0, 1)
    stored = np.zeros((6, 2))
    stored[:, 0] = tri_uvs.reshape(-1, 2)[:, 0] / 8.0
    stored[:, 1] = (1.0 - tri_uvs.reshape(-1, 2)[:, 1]) / 8.0
    uv16 = np.clip(np.round(stored * 65535.0), 0, 65535).astype(np.uint16)
    normals_uvs = np.concatenate([synthetic.encode_octahedral(normals), uv16], axis=1).astype(np.uint16)
    base, values = level_coded_texture(texture_size)
    textures = [base, constant_texture(texture_size, 131, 0), constant_texture(texture_size, 141)]
    # an unused emission material first, so that the quad does not emit
    scene = formats.Scene(["_emission", "quad"], 2, factor, summand, positions, normals_uvs, np.array([1, 1], np.uint8),
                          [constant_texture(4, 132), constant_texture(4, 131), constant_texture(4, 141)] + textures)
    return scene, values


def recovered_lod(albedo_red, values):
    """Inverts the level coding: the (fractional) level whose colour an albedo is."""
    linear = srgb_to_linear(np.array(values, np.float64))
    albedo_red = np.asarray(albedo_red, np.float64)
    lod = np.zeros_like(albedo_red)
    for m in range(len(values) - 1):
        inside = (albedo_red >= linear[m] - 1e-7) & (albedo_red <= linear[m + 1] + 1e-7)
        lod[inside] = m + (albedo_red[inside] - linear[m]) / (linear[m + 1] - linear[m])
    return lod


def test_level_of_detail_rule_closed_form():
    size, edge, repeats = 64, 2.0, 1.0
    scene, values = textured_quad(size, repeats, edge)
    o_scene, e_scene = oracle.OracleScene(scene), emu.EmuScene(scene)
    constants = oracle.write_constants(synthetic.look_at_spec((1, 1, 3), (0, 0, 0)), 32, 32, scene, np.zeros((0, 4), np.float32))
    texel = edge / (repeats * size)  # world size of a level-0 texel
    footprints = np.array([0.0, 0.5 * texel, texel, 1.5 * texel, 2 * texel, 3 * texel, 4 * texel, 8 * texel, 16 * texel, 1000 * texel], np.float32)
    n = len(footprints)
    tri = np.zeros(n, np.int32)
    barys = np.tile([0.3, 0.3], (n, 1)).astype(np.float32)
    front = np.ones(n, np.uint8)
    straight = np.tile([0.0, 0.0, 1.0], (n, 1)).astype(np.float32)
    a = o_scene.shading_data(constants, tri, barys, front, straight, footprints=footprints)
    expected = np.clip(np.log2(np.maximum(footprints.astype(np.float64) / texel, 1e-30)), 0, len(values) - 1)
    np.testing.assert_allclose(recovered_lod(a[:, 13], values), expected, atol=0.02)
    # Grazing directions: no change down to a cosine of 1/16, beyond that the footprint stretches by 1 / (16 cosine)
    for cosine, extra in ((0.5, 0.0), (0.07, 0.0), (1 / 32, 1.0), (1 / 64, 2.0)):
        direction = np.tile([np.sqrt(1 - cosine * cosine), 0.0, cosine], (n, 1)).astype(np.float32)
        g = o_scene.shading_data(constants, tri, barys, front, direction, footprints=footprints)
        grazing = np.clip(np.log2(np.maximum(footprints.astype(np.float64) / texel, 1e-30)) + extra, 0, len(values) - 1)
        np.testing.assert_allclose(recovered_lod(g[:, 13], values)[1:], grazing[1:], atol=0.03)
    # The kernel bodies agree bit for bit
    b = e_scene.shading_data(constants, tri, barys, front, straight, footprints=footprints)
    assert (bits(a) == bits(b)).all()


def hit_points(scene, tri, barys):
    """World positions of hits from triangle indices and barycentrics (float64, from the quantised vertices)."""
    q = scene.positions.astype(np.uint64)
    x = q[:, 0] & 0x1fffff
    y = ((q[:, 0] >> 21) & 0x7ff) | ((q[:, 1] & 0x3ff) << 11)
    z = (q[:, 1] >> 10) & 0x1fffff
    verts = np.stack([x, y, z], -1).astype(np.float64) * scene.dequantization_factor.astype(np.float64) + scene.dequantization_summand.astype(np.float64)
    verts = verts.reshape(-1, 3, 3)[tri]
    u, v = barys[..., 0:1].astype(np.float64), barys[..., 1:2].astype(np.float64)
    return verts[..., 0, :] * (1 - u - v) + verts[..., 1, :] * u + verts[..., 2, :] * v


@pytest.mark.parametrize("camera_type", [0, 1])
def test_primary_cone_is_one_pixel(camera_type):
    """Perspective and orthographic camera looking straight down at the quad from distance 3: the footprint that
    the cone of a primary ray has at its hit is the spacing of neighbouring pixels' hits (measured from the hits
    themselves: the Gaussian jitter has zero mean, so the mean spacing along a row is one pixel)."""
    size, edge, extent = 64, 2.0, 192
    scene, values = textured_quad(size, 1.0, edge)
    o_scene = oracle.OracleScene(scene)
    spec = synthetic.look_at_spec((1, 1, 3), (0, 0, 0), camera_type=camera_type)
    spec["height"] = 3.0
    constants = oracle.write_constants(spec, extent, extent, scene, np.zeros((0, 4), np.float32))
    tri, barys, front = o_scene.primary_hits(constants, 7)
    rows = slice(extent // 2 - 20, extent // 2 + 20)
    assert (tri[rows, rows] >= 0).all()
    points = hit_points(scene, tri[rows, rows], barys[rows, rows])
    spacing = np.abs(points[:, -1, 0] - points[:, 0, 0]).mean() / (points.shape[1] - 1)
    widths, spreads = o_scene.primary_cones(constants, 7)
    distance = 3.0 - (0.0 if camera_type else float(np.ravel(spec["near"])[0]))
    footprint = widths[rows, rows].astype(np.float64) + spreads[rows, rows].astype(np.float64) * distance
    # 8 significant bits travel with the ray (cut, not rounded: up to 0.8 % low); the jitter leaves about 2 % of noise in the spacing
    assert abs(footprint.mean() / spacing - 1.0) < 0.04
    if camera_type == 1:
        assert (spreads == 0).all()
    else:
        # Pixels at the edge of a perspective image subtend smaller angles than those at its centre
        assert spreads[extent // 2, 2] < 0.8 * spreads[extent // 2, extent // 2]
    e_widths, e_spreads = emu.EmuScene(scene).primary_cones(constants, 7)
    assert (bits(widths) == bits(e_widths)).all() and (bits(spreads) == bits(e_spreads)).all()


def minification_statistics(scene, evaluate, hits):
    """Spread and mean of the diffuse albedo of one material over many hits, per footprint."""
    tri, barys, front, out_dirs = hits
    keep = scene.materials[tri] == 1
    tri, barys, front, out_dirs = tri[keep], barys[keep], front[keep], out_dirs[keep]
    assert len(tri) > 200
    result = []
    for footprint in (0.0, 0.05, 0.2, 1.0, 100.0):
        data = evaluate(tri, barys, front, out_dirs, np.full(len(tri), footprint, np.float32))
        result.append((data[:, 13].std(), data[:, 13].mean()))
    return result


def check_minification(statistics):
    spreads = [s for s, _ in statistics]
    means = [m for _, m in statistics]
    # The spread shrinks as the footprint grows, to a fraction of what single texels show
    assert all(later <= earlier * 1.05 for earlier, later in zip(spreads, spreads[1:]))
    assert spreads[-1] < 0.3 * spreads[0]
    # while the mean stays (filtering happens on sRGB-decoded values, compression adds a little)
    assert max(means) - min(means) < 0.05


def test_minified_texture_converges_to_its_mean_without_aliasing():
    """A white-noise texture with filtered mip levels: through a large footprint every lookup is close to the
    texture's mean, while level 0 alone scatters over the whole range of the texels."""
    rng = np.random.default_rng(3)
    scene = synthetic.triangle_soup(seed=5, triangle_count=300, material_count=2, texture_size=64, filtered_mips=True)
    o_scene = oracle.OracleScene(scene)
    constants = oracle.write_constants(synthetic.look_at_spec((0, 9, 0), (np.pi / 2, 0, 0)), 16, 16, scene, np.zeros((0, 4), np.float32))
    o, d = rng.uniform(-6, 6, (6000, 3)).astype(np.float32), rng.normal(size=(6000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tri, tuv, front = o_scene.trace_rays(o, d)
    hit = tri >= 0
    hits = (tri[hit], tuv[hit, 1:], front[hit], -d[hit])
    check_minification(minification_statistics(scene, lambda t, b, f, o_, fp: o_scene.shading_data(constants, t, b, f, o_, footprints=fp), hits))


@pytest.mark.parametrize("strategy", [0, 2, 3])
def test_render_with_mip_chains_exact(strategy):
    """Kernel bodies against the oracle on a scene whose textures have full mip chains and whose paths reach
    coarse levels (small image, long rays, rough bounces)."""
    scene = synthetic.triangle_soup(seed=8, triangle_count=1500, material_count=4, texture_size=64)
    lights = synthetic.random_lights(2, 5)
    o_scene, e_scene = oracle.OracleScene(scene), emu.EmuScene(scene)
    e_scene.set_lights(lights)
    constants = oracle.write_constants(synthetic.look_at_spec((0, 9, 0), (np.pi / 2, 0, 0)), 40, 24, scene, lights)
    a, rays_a = o_scene.render(constants, lights, strategy=strategy, path_length=4, first_frame_index=11, frame_count=2, threads=4)
    b, rays_b = e_scene.render(constants, strategy=strategy, path_length=4, first_frame_index=11, frame_count=2)
    assert rays_a == rays_b
    assert ((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all()
