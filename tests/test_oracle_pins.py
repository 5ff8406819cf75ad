"""Pins for the CPU oracle: known answers, facts from the reference's data files, internal consistency.

The reference ships no tests or golden vectors, so these are the pins that exist: the published PCG2D algorithm
(JCGT 9(3), 2020) restated independently, values parsed from the reference's own data files, library functions
(scipy/numpy) for the elementary functions, and agreement of the four unbiased estimators.
"""
import os

import numpy as np
import pytest
import scipy.special

import oracle
from oracle import formats
from tests.paths import DATA


def pcg2d_python(x, y):
    m = 0xffffffff
    x = (1664525 * x + 1013904223) & m
    y = (1664525 * y + 1013904223) & m
    x = (x + 1664525 * y) & m
    y = (y + 1664525 * x) & m
    x ^= x >> 16
    y ^= y >> 16
    x = (x + 1664525 * y) & m
    y = (y + 1664525 * x) & m
    x ^= x >> 16
    y ^= y >> 16
    return x, y


def test_pcg2d_known_answers():
    randoms, seeds = oracle.rng_stream(0, 0, 0, 1)
    assert (int(seeds[0, 0]), int(seeds[0, 1])) == (0x802ed074, 0xe6ee372f)
    assert randoms[0, 0] == np.float32(0.5007143) and randoms[0, 1] == np.float32(0.90207237)
    _, seeds = oracle.rng_stream(255, 255, 2696, 1)
    assert (int(seeds[0, 0]), int(seeds[0, 1])) == (0x97f0d66d, 0xc37e4336)


def test_pcg2d_against_independent_restatement():
    rng = np.random.default_rng(7)
    for _ in range(50):
        x, y, f = (int(v) for v in rng.integers(0, 1 << 16, 3))
        randoms, seeds = oracle.rng_stream(x, y, f, 5)
        sx, sy = x ^ ((f << 16) & 0xffffffff), y ^ (((f + 237) << 16) & 0xffffffff)
        for i in range(5):
            sx, sy = pcg2d_python(sx, sy)
            assert (int(seeds[i, 0]), int(seeds[i, 1])) == (sx, sy)
            # uint -> float rounds to nearest even, then an exact scaling by 2^-32; 1.0 is reachable
            assert randoms[i, 0] == np.float32(sx) * np.float32(2.0 ** -32)
            assert randoms[i, 1] == np.float32(sy) * np.float32(2.0 ** -32)


def test_frame_index_shift_drops_high_bits():
    a, _ = oracle.rng_stream(3, 4, 5, 2)
    b, _ = oracle.rng_stream(3, 4, 5 + (1 << 16), 2)
    assert (a == b).all()


def test_cornell_box_file_facts(cornell, cornell_lights, cornell_spec):
    assert os.path.getsize(os.path.join(DATA, "cornell_box.vks")) == 64300
    assert cornell.material_names == ["_emission", "green", "red", "white"]
    assert cornell.triangle_count == 1310
    assert np.bincount(cornell.materials).tolist() == [1280, 2, 2, 26]
    assert cornell.emission_material_index == 0
    assert len(cornell.textures) == 12
    assert [t.format for t in cornell.textures[:3]] == [132, 131, 141]
    assert cornell.textures[0].payload.tobytes().hex() == "7ce7bce6aaaaaaaa"
    assert cornell_lights.shape == (1, 4)
    np.testing.assert_allclose(cornell_lights[0], [0.278, -0.2796, 0.4788, 0.07], rtol=1e-4)
    spec = cornell_spec[0]
    assert spec["scene_file"] == 1 and spec["frame_index"] == 2695 and spec["type"] == 0
    np.testing.assert_allclose(spec["position"], [0.278, 0.379, 0.2744], rtol=1e-6)
    np.testing.assert_allclose(spec["angles"], [np.pi / 2, 0, 0], atol=1e-6)
    assert spec["fov"] == np.float32(np.pi * float(np.float32(0.4)))
    assert spec["emission_material_strength"] == 20.0 and spec["sky_strength"] == 0.0


def _dequantize(scene):
    p = scene.positions.astype(np.uint64)
    x = (p[:, 0] & 0x1fffff).astype(np.float64)
    y = ((p[:, 0] >> 21) | ((p[:, 1] & 0x3ff) << 11)).astype(np.float64)
    z = ((p[:, 1] >> 10) & 0x1fffff).astype(np.float64)
    return (np.stack([x, y, z], 1) * scene.dequantization_factor + scene.dequantization_summand).reshape(-1, 3, 3)


def test_emission_mesh_is_the_light_sphere(cornell, cornell_lights):
    verts = _dequantize(cornell)[cornell.materials == 0].reshape(-1, 3)
    radius = np.linalg.norm(verts - cornell_lights[0, :3], axis=1)
    assert np.abs(radius - cornell_lights[0, 3]).max() < 2e-4


def test_front_face_convention_pin(cornell, cornell_lights, cornell_spec, oracle_cornell):
    """Every stored vertex normal agrees with cross(v1 - v0, v2 - v0); a camera inside the lit room must see front faces."""
    tris = _dequantize(cornell)
    geometric = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
    o = cornell.normals_uvs[:, :2].astype(np.float64) / 65535 * 2 * (65534 / 65535) - (32768 / 65535) * 2 * (65534 / 65535)
    nz = 1 - np.abs(o[:, 0]) - np.abs(o[:, 1])
    n = np.stack([o[:, 0], o[:, 1], nz], 1)
    fold = nz < 0
    n[fold, 0] = (1 - np.abs(o[fold, 1])) * np.where(o[fold, 0] >= 0, 1, -1)
    n[fold, 1] = (1 - np.abs(o[fold, 0])) * np.where(o[fold, 1] >= 0, 1, -1)
    stored = n.reshape(-1, 3, 3).mean(axis=1)
    assert ((geometric * stored).sum(axis=1) > 0).all()
    constants = oracle.write_constants(cornell_spec, 64, 64, cornell, cornell_lights)
    triangles, _, front = oracle_cornell.primary_hits(constants, 2696)
    assert front[triangles >= 0].all()


def test_elementary_functions():
    xs = np.linspace(-4.0, 7.0, 4001).astype(np.float32)
    for x in xs:
        s, c = oracle.sincos(float(x))
        assert abs(s - np.sin(np.float64(x))) < 3e-7 and abs(c - np.cos(np.float64(x))) < 3e-7
    for x in np.concatenate([np.geomspace(1e-37, 1e30, 500), [1.0, 0.5, 2.0]]).astype(np.float32):
        assert abs(oracle.log(float(x)) - np.log(np.float64(x))) <= 2e-7 * max(1.0, abs(np.log(np.float64(x))))
    for x in np.linspace(-0.9999, 0.9999, 801).astype(np.float32):
        expected = scipy.special.erfinv(np.float64(x))
        # 1 - x * x cancels in single precision near the ends (that is the reference's formula), hence the looser bound
        tolerance = 2e-6 if abs(x) <= 0.99 else 1e-4
        assert abs(oracle.erfinv(float(x)) - expected) <= tolerance * max(1.0, abs(expected))


def test_invert_mat4_and_camera(cornell, cornell_lights, cornell_spec):
    rng = np.random.default_rng(3)
    for _ in range(20):
        m = rng.normal(size=(4, 4)).astype(np.float32)
        inv = oracle.invert_mat4(m)
        np.testing.assert_allclose(inv.astype(np.float64) @ m.astype(np.float64), np.eye(4), atol=2e-3)
    constants = oracle.write_constants(cornell_spec, 512, 512, cornell, cornell_lights)
    w2p = np.array(constants.world_to_projection_space, np.float64).reshape(4, 4)
    p2w = np.array(constants.projection_to_world_space, np.float64).reshape(4, 4)
    np.testing.assert_allclose(w2p @ p2w, np.eye(4), atol=1e-5)
    assert constants.accum_frame_count == 1 and constants.frame_index == 2695
    assert tuple(constants.viewport_size) == (512.0, 512.0)
    assert tuple(constants.emission_material_radiance) == (20.0, np.float32(0.9) * np.float32(20.0), np.float32(0.6) * np.float32(20.0))
    # the camera looks along -y from y = 0.379: the hemispherical camera normal is row 2 of world to view
    np.testing.assert_allclose(tuple(constants.hemispherical_camera_normal), (0, 1, 0), atol=1e-6)


# ---- block decoding against an independent numpy decoder ----

def bc1_texel(block, px, py):
    c0, c1 = int.from_bytes(block[0:2], "little"), int.from_bytes(block[2:4], "little")
    bits = int.from_bytes(block[4:8], "little")
    code = (bits >> (2 * (4 * py + px))) & 3

    def expand(c):
        r, g, b = (c >> 11) & 31, (c >> 5) & 63, c & 31
        return np.array([(r << 3) | (r >> 2), (g << 2) | (g >> 4), (b << 3) | (b >> 2)])
    e0, e1 = expand(c0), expand(c1)
    if code == 0:
        return e0
    if code == 1:
        return e1
    if c0 > c1:
        return (2 * e0 + e1) // 3 if code == 2 else (e0 + 2 * e1) // 3
    return (e0 + e1) // 2 if code == 2 else np.zeros(3, int)


def bc4_texel(block, px, py):
    a0, a1 = block[0], block[1]
    bits = int.from_bytes(block[2:8], "little")
    code = (bits >> (3 * (4 * py + px))) & 7
    if a0 > a1:
        palette = [a0, a1] + [((7 - i) * a0 + i * a1) // 7 for i in range(1, 7)]
    else:
        palette = [a0, a1] + [((5 - i) * a0 + i * a1) // 5 for i in range(1, 5)] + [0, 255]
    return palette[code]


def srgb_to_linear(c):
    c = c / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


@pytest.mark.parametrize("fmt", [131, 132, 141])
def test_block_decode_at_texel_centers(fmt):
    rng = np.random.default_rng(fmt)
    size = 16
    block_bytes = 16 if fmt == 141 else 8
    payload = rng.integers(0, 256, (size // 4) * (size // 4) * block_bytes, dtype=np.uint8)
    texture = formats.Texture(size, size, fmt, 1, [(size, size, len(payload))], np.array([0], np.uint64), payload)
    scene = formats.Scene(["m"], 0, np.ones(3, np.float32), np.zeros(3, np.float32), np.zeros((0, 2), np.uint32), np.zeros((0, 4), np.uint16), np.zeros(0, np.uint8), [texture] * 3)
    s = oracle.OracleScene(scene)
    ys, xs = np.mgrid[0:size, 0:size]
    uvs = np.stack([(xs + 0.5) / size, (ys + 0.5) / size], -1).reshape(-1, 2)
    got = s.sample_texture(0, uvs).reshape(size, size, 3)
    for y in range(size):
        for x in range(size):
            block = bytes(payload[((y // 4) * (size // 4) + x // 4) * block_bytes:][:block_bytes])
            if fmt == 141:
                expected = np.array([bc4_texel(block[:8], x & 3, y & 3), bc4_texel(block[8:], x & 3, y & 3), 0]) / 255.0
            else:
                rgb = bc1_texel(block, x & 3, y & 3)
                expected = srgb_to_linear(rgb) if fmt == 132 else rgb / 255.0
            np.testing.assert_allclose(got[y, x], expected, atol=1e-6)


def test_bilinear_filtering_and_repeat():
    # a 4x4 BC1 texture with two colours: left half code 0, right half code 1
    c0, c1 = 0xf800, 0x001f
    codes = 0
    for py in range(4):
        for px in range(4):
            codes |= (0 if px < 2 else 1) << (2 * (4 * py + px))
    payload = np.frombuffer(c0.to_bytes(2, "little") + c1.to_bytes(2, "little") + codes.to_bytes(4, "little"), np.uint8)
    texture = formats.Texture(4, 4, 131, 1, [(4, 4, 8)], np.array([0], np.uint64), payload)
    scene = formats.Scene(["m"], 0, np.ones(3, np.float32), np.zeros(3, np.float32), np.zeros((0, 2), np.uint32), np.zeros((0, 4), np.uint16), np.zeros(0, np.uint8), [texture] * 3)
    s = oracle.OracleScene(scene)
    # halfway between texel 1 (red) and texel 2 (blue)
    np.testing.assert_allclose(s.sample_texture(0, [[0.5, 0.5]])[0], [0.5, 0, 0.5], atol=1e-6)
    # REPEAT: u = 0 lies halfway between texel 3 (blue) and texel 0 (red); u + 1 and u - 3 are the same place
    np.testing.assert_allclose(s.sample_texture(0, [[0.0, 0.5]])[0], [0.5, 0, 0.5], atol=1e-6)
    a = s.sample_texture(0, [[0.3, 0.2], [1.3, 0.2], [-2.7, 1.2]])
    np.testing.assert_allclose(a[0], a[1], atol=1e-6)
    np.testing.assert_allclose(a[0], a[2], atol=2e-6)


def test_path_length_semantics(cornell, cornell_lights, cornell_spec, oracle_cornell):
    constants = oracle.write_constants(cornell_spec, 32, 32, cornell, cornell_lights)
    black, rays = oracle_cornell.render(constants, cornell_lights, path_length=0, first_frame_index=5, frame_count=2)
    assert rays == 0 and (black[..., :3] == 0).all() and (black[..., 3] == 1).all()
    emission_only, rays = oracle_cornell.render(constants, cornell_lights, path_length=1, first_frame_index=5, frame_count=1)
    assert rays == 32 * 32
    colours = np.unique(emission_only[..., :3].reshape(-1, 3), axis=0)
    assert len(colours) == 2 and (colours[0] == 0).all()
    np.testing.assert_allclose(colours[1], [20.0, 18.0, 12.0], rtol=1e-6)


def test_oracle_regression_vectors():
    """The oracle against renders it produced when they were committed (tests/golden/make_vectors.py): not a pin by
    the reference, but one that a change altering oracle and HIP library alike cannot pass unnoticed."""
    import importlib.util
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_vectors", os.path.join(here, "make_vectors.py"))
    make_vectors = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(make_vectors)
    expected_rays = dict(line.split() for line in open(os.path.join(here, "vectors", "ray_counts.txt")))
    for name, image, rays in make_vectors.render_cases():
        expected = np.load(os.path.join(here, "vectors", "cornell_%s.npy" % name))
        assert rays == int(expected_rays[name]), name
        assert (image.view(np.uint32) == expected.view(np.uint32)).all(), name
        assert image[..., :3].max() > 0


@pytest.mark.parametrize("tracer", ["oracle", "kernel bodies"])
def test_ray_queries_against_double_precision(tracer):
    """What a ray query returns (pathtrace.frag.glsl:184-197 leaves it to Vulkan): the closest intersection with
    t > 1e-3, barycentric coordinates of the second and third vertex, front face where the triangle appears
    counter-clockwise... as pinned on the Cornell box, i.e. dot(cross(v1 - v0, v2 - v0), direction) < 0. Checked against
    a ray/triangle intersection in double precision over all triangles (Moeller-Trumbore, numpy), which shares no
    code with the oracle's watertight test, on the rays whose answer does not hinge on the last bits: the hit is not
    within 1e-4 of an edge, no other triangle comes within 1e-4 of being hit first."""
    from tests import emu, synthetic
    scene = synthetic.triangle_soup(seed=91, triangle_count=400, material_count=2, size=1.2, texture_size=4)
    words = scene.positions.reshape(-1, 3, 2).astype(np.uint64)
    quantized = np.stack([words[..., 0] & 0x1fffff, (words[..., 0] >> 21) | ((words[..., 1] & 0x3ff) << 11), (words[..., 1] >> 10) & 0x1fffff], axis=2)
    vertices = quantized.astype(np.float64) * scene.dequantization_factor.astype(np.float64) + scene.dequantization_summand.astype(np.float64)
    rng = np.random.default_rng(92)
    count = 6000
    origins = rng.uniform(-6.0, 6.0, (count, 3)).astype(np.float32)
    directions = rng.normal(size=(count, 3))
    directions = (directions / np.linalg.norm(directions, axis=1, keepdims=True)).astype(np.float32)
    o, d = origins.astype(np.float64)[:, None, :], directions.astype(np.float64)[:, None, :]
    v0, e1, e2 = vertices[None, :, 0], (vertices[:, 1] - vertices[:, 0])[None], (vertices[:, 2] - vertices[:, 0])[None]
    p = np.cross(d, e2)
    det = (e1 * p).sum(axis=2)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / det
        tv = o - v0
        u = (tv * p).sum(axis=2) * inv
        q = np.cross(tv, e1)
        v = (d * q).sum(axis=2) * inv
        t = (e2 * q).sum(axis=2) * inv
    inside = np.minimum(np.minimum(u, v), 1.0 - u - v)
    margin = 1e-4
    ahead = np.isfinite(t) & (t > 1e-3 * (1.0 + margin))
    sure = ahead & (inside > margin)
    maybe = np.isfinite(t) & (t > 1e-3 * (1.0 - margin)) & (inside > -margin)
    t_sure = np.where(sure, t, np.inf)
    best = t_sure.argmin(axis=1)
    best_t = t_sure.min(axis=1)
    hit = np.isfinite(best_t)
    # Nothing doubtful may come first, and no second sure hit may be as near as the first
    others = np.where(maybe, t, np.inf)
    others[np.arange(count), np.where(hit, best, 0)] = np.where(hit, np.inf, others[np.arange(count), 0])
    robust_hit = hit & (others.min(axis=1) > best_t * (1.0 + margin))
    robust_miss = ~maybe.any(axis=1)
    assert robust_hit.sum() > 0.25 * count and robust_miss.sum() > 0.05 * count
    tracer_scene = oracle.OracleScene(scene) if tracer == "oracle" else emu.EmuScene(scene)
    result = tracer_scene.trace_rays(origins, directions)
    triangles, tuv, front = result[0], result[1], result[2]
    assert (triangles[robust_miss] == -1).all()
    rays = np.flatnonzero(robust_hit)
    assert (triangles[rays] == best[rays]).all()
    # Single precision on coordinates of up to eight units: a few 1e-5 in distance, less in the coordinates
    np.testing.assert_allclose(tuv[rays, 0], best_t[rays], rtol=2e-5, atol=5e-5)
    np.testing.assert_allclose(tuv[rays, 1], u[rays, best[rays]], atol=1e-4)
    np.testing.assert_allclose(tuv[rays, 2], v[rays, best[rays]], atol=1e-4)
    # det = dot(v1 - v0, cross(direction, v2 - v0)) = -dot(cross(v1 - v0, v2 - v0), direction)
    assert (front[rays].astype(bool) == (det[rays, best[rays]] > 0.0)).all()


@pytest.mark.parametrize("restatement", ["oracle", "kernel bodies"])
def test_shading_data_against_a_restatement_in_double_precision(restatement):
    """get_shading_data() (shading_data.glsl:51-103) with dequantize_position() and dequantize_normal()
    (mesh_quantization.glsl:4-31), written out again in numpy and double precision for random triangles with random
    vertex normals and texture coordinates, materials of one colour each (so that texture filtering, which has its
    own tests, does not enter: the decoded texel values are read through the test hook) with tilted normal maps,
    front and back faces, and directions towards the camera on both sides of the surface: position, shading normal
    (tangent frame from the texture coordinates, the bend towards the camera), albedo, Fresnel term, roughness and
    emission agree to single precision."""
    from tests import emu, synthetic
    soup = synthetic.triangle_soup(seed=131, triangle_count=300, material_count=4, size=0.8, texture_size=4)
    values = [((0.0, 0.0, 0.0), (1.0, 1.0, 0.0), (0.5, 0.5)), ((0.8, 0.3, 0.2), (0.3, 0.7, 0.0), (0.5, 0.5)),
              ((0.2, 0.6, 0.9), (0.9, 0.3, 1.0), (0.8, 0.35)), ((0.5, 0.5, 0.1), (0.5, 0.05, 0.4), (0.2, 0.7))]
    textures = []
    for base, specular, normal in values:
        textures += [synthetic.constant_texture(base, 132), synthetic.constant_texture(specular, 131), synthetic.constant_texture(normal, 141)]
    scene = formats.Scene(soup.material_names, soup.triangle_count, soup.dequantization_factor, soup.dequantization_summand,
                          soup.positions, soup.normals_uvs, soup.materials, textures)
    checker = oracle.OracleScene(scene)
    subject = checker if restatement == "oracle" else emu.EmuScene(scene)
    rng = np.random.default_rng(132)
    count = 2000
    triangles = rng.integers(0, scene.triangle_count, count)
    barys = rng.dirichlet((1.0, 1.0, 1.0), count)[:, 1:]
    front = rng.integers(0, 2, count).astype(np.uint8)
    out_dirs = rng.normal(size=(count, 3))
    out_dirs /= np.linalg.norm(out_dirs, axis=1, keepdims=True)
    spec = synthetic.look_at_spec((0.5, 7.0, 0.3), (1.3, 0.0, 0.4), emission_strength=3.0)
    constants = oracle.write_constants(spec, 16, 16, scene, np.zeros((0, 4), np.float32))
    got = subject.shading_data(constants, triangles, barys.astype(np.float32), front, out_dirs.astype(np.float32)).astype(np.float64)

    # ---- the restatement ----
    barys = barys.astype(np.float32).astype(np.float64)
    out_dirs = out_dirs.astype(np.float32).astype(np.float64)
    weights = np.stack([1.0 - barys[:, 0] - barys[:, 1], barys[:, 0], barys[:, 1]], axis=1)
    words = scene.positions.reshape(-1, 3, 2).astype(np.uint64)[triangles]
    quantized = np.stack([words[..., 0] & 0x1fffff, (words[..., 0] >> 21) | ((words[..., 1] & 0x3ff) << 11), (words[..., 1] >> 10) & 0x1fffff], axis=2)
    poss = quantized.astype(np.float64) * scene.dequantization_factor.astype(np.float64) + scene.dequantization_summand.astype(np.float64)
    attributes = scene.normals_uvs.reshape(-1, 3, 4)[triangles].astype(np.float64) / 65535.0
    factor = 2.0 * (65534.0 / 65535.0)
    octahedral = attributes[..., 0:2] * factor - (32768.0 / 65535.0) * factor
    normals = np.concatenate([octahedral, 1.0 - np.abs(octahedral).sum(axis=2, keepdims=True)], axis=2)
    folded = (1.0 - np.abs(normals[..., [1, 0]])) * np.where(octahedral >= 0.0, 1.0, -1.0)
    normals[..., 0:2] = np.where(normals[..., 2:3] < 0.0, folded, normals[..., 0:2])
    normals /= np.linalg.norm(normals, axis=2, keepdims=True)
    tex_coords = attributes[..., 2:4] * np.array([8.0, -8.0]) + np.array([0.0, 1.0])
    pos = (weights[:, :, None] * poss).sum(axis=1)
    normal_geo = (weights[:, :, None] * normals).sum(axis=1)
    normal_geo /= np.linalg.norm(normal_geo, axis=1, keepdims=True)
    material = scene.materials[triangles].astype(np.int64)
    texel = {m: [checker.sample_texture(3 * m + i, [[0.3, 0.6]])[0].astype(np.float64) for i in range(3)] for m in range(4)}
    base = np.array([texel[m][0] for m in material])
    specular = np.array([texel[m][1] for m in material])
    normal_tex = np.array([texel[m][2][:2] for m in material])
    local = np.zeros((count, 3))
    local[:, 0:2] = normal_tex * 2.0 - 1.0
    local[:, 2] = np.sqrt(np.maximum(0.0, 1.0 - local[:, 0] ** 2 - local[:, 1] ** 2))
    edge_1, edge_2 = tex_coords[:, 1] - tex_coords[:, 0], tex_coords[:, 2] - tex_coords[:, 0]
    pre_0 = np.cross(normal_geo, poss[:, 1] - poss[:, 0])
    pre_1 = np.cross(normal_geo, poss[:, 0] - poss[:, 2])
    # mat2(a, b)[0] is the column a: tex_edges[0][0] = edge_1.x, tex_edges[1][0] = edge_2.x
    tangent_0 = pre_1 * edge_1[:, 0:1] + pre_0 * edge_2[:, 0:1]
    tangent_1 = pre_1 * edge_1[:, 1:2] + pre_0 * edge_2[:, 1:2]
    mean_length = np.sqrt(0.5 * ((tangent_0 ** 2).sum(axis=1) + (tangent_1 ** 2).sum(axis=1)))
    local[:, 2] *= np.maximum(1.0e-8, mean_length)
    normal = tangent_0 * local[:, 0:1] + tangent_1 * local[:, 1:2] + normal_geo * local[:, 2:3]
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    normal = np.where(front[:, None] != 0, normal, -normal)
    offset = np.maximum(0.0, 1.0e-3 - (normal * out_dirs).sum(axis=1))
    normal = normal + offset[:, None] * out_dirs
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    metalicity = specular[:, 2:3]
    expected = np.zeros((count, 20))
    expected[:, 0:3] = pos
    expected[:, 3:6] = normal
    expected[:, 6:9] = out_dirs
    expected[:, 9] = (normal * out_dirs).sum(axis=1)
    expected[:, 10:13] = np.where(material[:, None] == scene.emission_material_index, np.array(constants.emission_material_radiance[:3]), 0.0)
    expected[:, 13:16] = base - metalicity * base
    expected[:, 16:19] = 0.02 * (1.0 - metalicity) + base * metalicity
    expected[:, 19] = np.maximum(0.006, specular[:, 1] ** 2)
    # Where the tangent frame nearly degenerates (texture coordinates of a triangle nearly collinear) the normal
    # hinges on the last bits of single precision; everywhere else it agrees to a few 1e-5
    stable = mean_length > 1e-3
    assert stable.mean() > 0.95 and (material == scene.emission_material_index).any() and (offset > 0.0).mean() > 0.3
    np.testing.assert_allclose(got[:, 0:3], expected[:, 0:3], atol=2e-5)
    np.testing.assert_allclose(got[stable, 3:6], expected[stable, 3:6], atol=2e-4)
    np.testing.assert_allclose(got[stable, 9], expected[stable, 9], atol=2e-4)
    np.testing.assert_allclose(got[:, 6:9], expected[:, 6:9], atol=1e-6)
    np.testing.assert_allclose(got[:, 10:20], expected[:, 10:20], atol=2e-6, rtol=2e-6)


def test_camera_rays_against_unprojection_in_double_precision():
    """The camera ray of a pixel (pathtrace.frag.glsl:368-380, camera_utilities.glsl:9-44: Gaussian jitter of 0.9
    pixels from the first random pair, origin on the near plane, direction from the minors of the world to
    projection space matrix) against plain unprojection in double precision: the point where the ray of every pixel
    meets a floor must be the point where the line through the unprojected near and far points of the jittered pixel
    position meets it. Pins the pixel to ray mapping (orientation of both axes, pixel centres, the jitter's scale)
    through the primary hits of the oracle."""
    from tests import synthetic
    scene = synthetic.furnace_scene(half_extent=200.0)
    spec = synthetic.look_at_spec((1.5, -6.0, 4.0), (1.0, 0.0, 0.35), sky_strength=1.0)
    width, height, frame = 48, 32, 77
    constants = oracle.write_constants(spec, width, height, scene, np.zeros((0, 4), np.float32))
    triangles, barys, front = oracle.OracleScene(scene).primary_hits(constants, frame)
    hit = triangles >= 0
    assert hit.mean() > 0.3
    words = scene.positions.reshape(-1, 3, 2).astype(np.uint64)
    quantized = np.stack([words[..., 0] & 0x1fffff, (words[..., 0] >> 21) | ((words[..., 1] & 0x3ff) << 11), (words[..., 1] >> 10) & 0x1fffff], axis=2)
    vertices = quantized.astype(np.float64) * scene.dequantization_factor.astype(np.float64) + scene.dequantization_summand.astype(np.float64)
    projection_to_world = np.linalg.inv(np.array(constants.world_to_projection_space, np.float64).reshape(4, 4))
    worst = 0.0
    for y in range(height):
        for x in range(width):
            if not hit[y, x]:
                continue
            u = oracle.rng_stream(x, y, frame, 1)[0][0].astype(np.float64)
            jitter = 0.9 * np.sqrt(2.0) * scipy.special.erfinv(2.0 * u - 1.0)
            ndc = 2.0 * (np.array([x + 0.5, y + 0.5]) + jitter) / np.array([width, height]) - 1.0
            near = projection_to_world @ np.array([ndc[0], ndc[1], 0.0, 1.0])
            far = projection_to_world @ np.array([ndc[0], ndc[1], 1.0, 1.0])
            near, far = near[:3] / near[3], far[:3] / far[3]
            direction = far - near
            expected = near - near[2] / direction[2] * direction
            b = barys[y, x].astype(np.float64)
            v = vertices[triangles[y, x]]
            got = (1.0 - b[0] - b[1]) * v[0] + b[0] * v[1] + b[1] * v[2]
            distance = np.linalg.norm(expected - near)
            worst = max(worst, np.linalg.norm(got - expected) / distance)
    # Single precision in the matrix inverse, the minors and the intersection: a few 1e-6 of the distance
    assert worst < 5e-5, worst


@pytest.mark.parametrize("camera_type", [2, 3])
def test_hemispherical_and_spherical_camera_rays(camera_type):
    """The rays of the hemispherical and the spherical camera (pathtrace.frag.glsl:381-386 with
    sample_hemisphere_spherical() at :48-53 and the frame of brdfs.glsl:126-134, which is the published branchless
    frame of Duff et al.), written out again in double precision: the direction from the camera position to the point
    where a pixel's ray meets a floor is the frame about the camera's normal applied to the spherical coordinates of
    the jittered pixel position."""
    from tests import synthetic
    scene = synthetic.furnace_scene(half_extent=500.0)
    spec = synthetic.look_at_spec((0.3, -0.2, 3.0), (np.pi - 0.4, 0.0, 0.3), sky_strength=1.0, camera_type=camera_type)
    width, height, frame = 40, 24, 5
    constants = oracle.write_constants(spec, width, height, scene, np.zeros((0, 4), np.float32))
    normal = np.array(constants.hemispherical_camera_normal[:3], np.float64)
    position = np.array(constants.camera_pos[:3], np.float64)
    assert normal[2] < -0.5 and abs(np.linalg.norm(normal) - 1.0) < 1e-6
    sign = np.copysign(1.0, normal[2])
    a = -1.0 / (sign + normal[2])
    b = normal[0] * normal[1] * a
    tangent_0 = np.array([1.0 + sign * normal[0] * normal[0] * a, sign * b, -sign * normal[0]])
    tangent_1 = np.array([b, sign + normal[1] * normal[1] * a, -normal[1]])
    triangles, barys, front = oracle.OracleScene(scene).primary_hits(constants, frame)
    hit = triangles >= 0
    assert hit.mean() > (0.5 if camera_type == 2 else 0.25)
    words = scene.positions.reshape(-1, 3, 2).astype(np.uint64)
    quantized = np.stack([words[..., 0] & 0x1fffff, (words[..., 0] >> 21) | ((words[..., 1] & 0x3ff) << 11), (words[..., 1] >> 10) & 0x1fffff], axis=2)
    vertices = quantized.astype(np.float64) * scene.dequantization_factor.astype(np.float64) + scene.dequantization_summand.astype(np.float64)
    worst = 0.0
    for y in range(height):
        for x in range(width):
            if not hit[y, x]:
                continue
            u = oracle.rng_stream(x, y, frame, 1)[0][0].astype(np.float64)
            jittered = np.array([x + 0.5, y + 0.5]) + 0.9 * np.sqrt(2.0) * scipy.special.erfinv(2.0 * u - 1.0)
            coordinates = jittered * np.array([1.0, 2.0 if camera_type == 3 else 1.0]) / np.array([width, height])
            azimuth, inclination = 2.0 * np.pi * coordinates[0] - np.pi, 0.5 * np.pi * coordinates[1]
            local = np.array([np.sin(inclination) * np.cos(azimuth), np.sin(inclination) * np.sin(azimuth), np.cos(inclination)])
            expected = tangent_0 * local[0] + tangent_1 * local[1] + normal * local[2]
            weights = barys[y, x].astype(np.float64)
            v = vertices[triangles[y, x]]
            got = (1.0 - weights[0] - weights[1]) * v[0] + weights[0] * v[1] + weights[1] * v[2] - position
            got /= np.linalg.norm(got)
            if expected[2] > -0.02:
                continue  # grazing the floor: the meeting point is too far away to tell directions apart
            worst = max(worst, np.linalg.norm(got - expected))
    assert worst < 5e-5, worst
