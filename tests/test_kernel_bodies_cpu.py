"""The per-thread bodies of the gfx950 kernels (BVH build steps, traversal, shading), executed on the host by
tests/emu, against the CPU oracle. Bit-exact: both sides use the same order of IEEE operations."""
import numpy as np
import pytest

import oracle
from tests import emu, synthetic


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def emu_cornell(cornell, cornell_lights):
    scene = emu.EmuScene(cornell)
    scene.set_lights(cornell_lights)
    return scene


@pytest.fixture(scope="module")
def soup():
    scene = synthetic.triangle_soup(seed=2, triangle_count=3000, material_count=4)
    lights = synthetic.random_lights(3, 6)
    e = emu.EmuScene(scene)
    e.set_lights(lights)
    return scene, lights, oracle.OracleScene(scene), e


def test_elementary_functions_bitwise():
    lib = emu.lib()
    import ctypes as C
    for x in np.linspace(-7, 7, 2001).astype(np.float32):
        s, c = C.c_float(), C.c_float()
        lib.emu_sincos(float(x), C.byref(s), C.byref(c))
        assert (s.value, c.value) == oracle.sincos(float(x))
    for x in np.geomspace(1e-37, 1e30, 1000).astype(np.float32):
        assert lib.emu_log(float(x)) == oracle.log(float(x))
    for x in np.linspace(-1, 1, 2001).astype(np.float32):
        a, b = lib.emu_erfinv(float(x)), oracle.erfinv(float(x))
        assert a == b or (np.isnan(a) and np.isnan(b))


def test_unorm_conversions_equal_ieee_division():
    """x / 255 and x / 65535 by reciprocal and one residual correction: the same bits as the division for every texel value."""
    assert emu.lib().emu_unorm_mismatches() == 0


def test_bvh_shape_cornell(emu_cornell):
    info = emu_cornell.bvh_info()
    assert 1310 / 24 <= info["node_count"] <= 1310
    assert info["depth"] <= 12


def random_rays(rng, n, lo, hi):
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def test_traversal_exact_cornell(oracle_cornell, emu_cornell):
    o, d = random_rays(np.random.default_rng(1), 50000, [0, -0.56, 0], [0.556, 0.4, 0.549])
    # axis-aligned and zero-component directions exercise the reciprocal clamping of the box test
    d[:300] = np.eye(3, dtype=np.float32)[np.arange(300) % 3] * np.where(np.arange(300) % 2, 1, -1)[:, None]
    a = oracle_cornell.trace_rays(o, d)
    b = emu_cornell.trace_rays(o, d)
    assert (a[0] == b[0]).all() and (bits(a[1]) == bits(b[1])).all() and (a[2] == b[2]).all()
    assert 0.3 < (a[0] >= 0).mean() < 0.9
    assert emu_cornell.bvh_info()["max_stack"] <= emu_cornell.bvh_info()["depth"]


def test_traversal_exact_soup(soup):
    scene, lights, o_scene, e_scene = soup
    o, d = random_rays(np.random.default_rng(4), 30000, [-6, -6, -6], [6, 6, 6])
    a = o_scene.trace_rays(o, d)
    b = e_scene.trace_rays(o, d)
    assert (a[0] == b[0]).all() and (bits(a[1]) == bits(b[1])).all() and (a[2] == b[2]).all()
    assert (a[0] >= 0).mean() > 0.5
    assert e_scene.bvh_info()["max_stack"] <= e_scene.bvh_info()["depth"]


def test_shading_data_exact_soup(soup):
    scene, lights, o_scene, e_scene = soup
    rng = np.random.default_rng(5)
    o, d = random_rays(rng, 20000, [-6, -6, -6], [6, 6, 6])
    tri, tuv, front = o_scene.trace_rays(o, d)
    hit = tri >= 0
    spec = synthetic.look_at_spec((0, 9, 0), (np.pi / 2, 0, 0))
    constants = oracle.write_constants(spec, 64, 64, scene, lights)
    a = o_scene.shading_data(constants, tri[hit], tuv[hit, 1:], front[hit], -d[hit])
    b = e_scene.shading_data(constants, tri[hit], tuv[hit, 1:], front[hit], -d[hit])
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    assert same.all()
    # textures really vary: many distinct albedos and roughness values
    assert len(np.unique(a[:, 13])) > 100 and len(np.unique(a[:, 19])) > 50


@pytest.mark.parametrize("strategy", [0, 1, 2, 3])
def test_render_exact_cornell(strategy, cornell, cornell_lights, cornell_spec, oracle_cornell, emu_cornell):
    constants = oracle.write_constants(cornell_spec, 48, 40, cornell, cornell_lights)
    a, rays_a = oracle_cornell.render(constants, cornell_lights, strategy=strategy, path_length=4, first_frame_index=2696, frame_count=3, threads=4)
    b, rays_b = emu_cornell.render(constants, strategy=strategy, path_length=4, first_frame_index=2696, frame_count=3)
    assert rays_a == rays_b
    assert ((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all()


@pytest.mark.parametrize("camera_type", [0, 1, 2, 3])
def test_render_exact_soup_all_cameras(camera_type, soup):
    scene, lights, o_scene, e_scene = soup
    spec = synthetic.look_at_spec((0.5, 7.0, 0.3), (1.3, 0.0, 0.4), camera_type=camera_type)
    constants = oracle.write_constants(spec, 40, 24, scene, lights)
    a, rays_a = o_scene.render(constants, lights, strategy=3, path_length=5, first_frame_index=77, frame_count=2, frame_stride=3, threads=4)
    b, rays_b = e_scene.render(constants, strategy=3, path_length=5, first_frame_index=77, frame_count=2, frame_stride=3)
    assert rays_a == rays_b
    assert ((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all()
    assert np.nanmax(a[..., :3]) > 0


def test_tiny_scenes():
    for count in (1, 2, 3, 4, 9):
        scene = synthetic.triangle_soup(seed=10 + count, triangle_count=count, material_count=2, extent=2.0, size=1.0, texture_size=4)
        o_scene, e_scene = oracle.OracleScene(scene), emu.EmuScene(scene)
        o, d = random_rays(np.random.default_rng(count), 4000, [-3, -3, -3], [3, 3, 3])
        a, b = o_scene.trace_rays(o, d), e_scene.trace_rays(o, d)
        assert (a[0] == b[0]).all() and (bits(a[1]) == bits(b[1])).all()
        assert (a[0] >= 0).any()


def test_duplicate_triangles_tie_break():
    """Coincident triangles give equal t: the lower index must win whatever the BVH looks like."""
    scene = synthetic.triangle_soup(seed=21, triangle_count=200, material_count=2)
    scene.positions = np.concatenate([scene.positions, scene.positions, scene.positions])
    scene.normals_uvs = np.concatenate([scene.normals_uvs] * 3)
    scene.materials = np.concatenate([scene.materials] * 3)
    scene.triangle_count = 600
    o_scene, e_scene = oracle.OracleScene(scene), emu.EmuScene(scene)
    o, d = random_rays(np.random.default_rng(2), 20000, [-6, -6, -6], [6, 6, 6])
    a, b = o_scene.trace_rays(o, d), e_scene.trace_rays(o, d)
    assert (a[0] == b[0]).all()
    assert (a[0][a[0] >= 0] < 200).all()


def test_tree_quality_against_sah_yardstick(tmp_path, monkeypatch):
    """The builder's tree (radix tree + optimal wide collapse) against a top-down binned-SAH tree over the same
    collapse: same hits, and at most 1.15 times the node visits per ray on a generated scene (without the treelet
    passes the radix tree needs 1.44 times)."""
    from oracle import formats
    from path_tracer_amd.tools import procedural
    root = str(tmp_path / "hall")
    procedural.ensure_scene("interior", root, triangles=40000, texture_size=16, material_count=8)
    scene = formats.load_scene(root + "/data/Arcade.vks", root + "/data/Arcade_textures")
    rng = np.random.default_rng(4)
    o = rng.uniform([0.5, 0.5, 0.2], [23.5, 15.5, 4.5], (20000, 3)).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    results = {}
    for builder in ("lbvh", "sah"):
        monkeypatch.setenv("PT_EMU_BUILDER", builder)
        tree = emu.EmuScene(scene)
        tri, tuv, front, visits = tree.trace_rays(o, d)
        results[builder] = (tri, bits(tuv), visits[0] / len(o), visits[1] / len(o), tree.bvh_info())
        tree.close()
    assert (results["lbvh"][0] == results["sah"][0]).all() and (results["lbvh"][1] == results["sah"][1]).all()
    assert (results["lbvh"][0] >= 0).mean() > 0.99  # a closed hall
    print("nodes per ray: builder %.2f, yardstick %.2f" % (results["lbvh"][2], results["sah"][2]))
    assert results["lbvh"][2] <= 1.15 * results["sah"][2], (results["lbvh"][2], results["sah"][2])
    assert results["lbvh"][3] <= 8.0  # one triangle per leaf child: few triangle tests per ray
    assert results["lbvh"][4]["node_count"] < scene.triangle_count / 3


@pytest.mark.parametrize("light_count", [1, 2, 5, 37, 300])
def test_render_exact_light_tree(light_count):
    """Next event estimation through the light tree: the kernel body against the oracle's restatement of the same
    definition, for light counts that are and are not powers of two (empty leaves) and exceed the reference's 32."""
    scene = synthetic.triangle_soup(seed=30 + light_count, triangle_count=1500, material_count=3)
    lights = synthetic.random_lights(40 + light_count, light_count)
    o_scene, e_scene = oracle.OracleScene(scene), emu.EmuScene(scene)
    e_scene.set_lights(lights)
    spec = synthetic.look_at_spec((0.5, 7.0, 0.3), (1.3, 0.0, 0.4))
    constants = oracle.write_constants(spec, 40, 24, scene, lights)
    a, rays_a = o_scene.render(constants, lights, strategy=3, path_length=4, first_frame_index=5, frame_count=2, threads=4, light_sampling=1)
    b, rays_b = e_scene.render(constants, strategy=4, path_length=4, first_frame_index=5, frame_count=2)
    assert rays_a == rays_b
    assert ((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all()
    # The tree changes which light a vertex samples, so the image differs from the loop over all lights
    c, _ = o_scene.render(constants, lights, strategy=3, path_length=4, first_frame_index=5, frame_count=2, threads=4)
    if light_count > 1:
        assert not (bits(a) == bits(c)).all()


@pytest.mark.parametrize("layout", synthetic.LIGHT_LAYOUTS)
def test_light_tree_layouts_against_oracle(layout, monkeypatch):
    """Which lights a ray meets is looked up in a grid over the lights; the walk of the light tree serves what the grid
    does not. For light sets that stress both (synthetic.light_layout), the kernel body with the grid, the kernel body
    with the walk alone and the oracle's own walk, one ray at a time, agree bit for bit."""
    lights = synthetic.light_layout(layout)
    scene = synthetic.triangle_soup(seed=31, triangle_count=1200, material_count=3)
    o_scene, e_scene = oracle.OracleScene(scene), emu.EmuScene(scene)
    e_scene.set_lights(lights)
    spec = synthetic.look_at_spec((0.5, 7.0, 0.3), (1.3, 0.0, 0.4))
    constants = oracle.write_constants(spec, 40, 24, scene, lights)
    a, rays_a = o_scene.render(constants, lights, strategy=3, path_length=4, first_frame_index=3, frame_count=2, threads=4, light_sampling=1)
    assert e_scene.light_grid_info()[3] >= len(lights)
    b, rays_b = e_scene.render(constants, strategy=4, path_length=4, first_frame_index=3, frame_count=2)
    monkeypatch.setenv("PT_EMU_NO_LIGHT_GRID", "1")
    c, rays_c = e_scene.render(constants, strategy=4, path_length=4, first_frame_index=3, frame_count=2)
    assert rays_a == rays_b == rays_c
    assert ((bits(b) == bits(c)) | (np.isnan(b) & np.isnan(c))).all()
    assert ((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all()
    assert b[..., :3].sum() > 0


def test_light_grid_and_tree_walk_find_what_a_loop_over_all_lights_finds():
    """The grid over the lights returns exactly the lights that pass the test for a ray meeting a light, and the
    densities through the grid, by the walk of the light tree and by a loop over all lights are the same bits: for rays
    aimed at lights (also tangentially) and anywhere, along axes and within axis planes, from among the lights and from
    up to a thousand times their extent away (beyond the grid's reach the walk serves), with directions of any length.
    Grid and boxes of the tree allow for the rounding of the test. And the selection probability kept from the descent
    equals the one recomputed from the root."""
    rng = np.random.default_rng(12)
    scene = synthetic.triangle_soup(seed=3, triangle_count=50, material_count=2)
    e_scene = emu.EmuScene(scene)

    def unit(v):
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)

    sets = {"uniform": synthetic.random_lights(5, 300),
            "one": np.array([[1.0, 2.0, 3.0, 0.4]], np.float32),
            "big and small": np.concatenate([synthetic.random_lights(6, 100), np.array([[0, 0, 60.0, 30.0], [3, 3, 3, 2.5]], np.float32)]),
            "flat": synthetic.random_lights(7, 200) * np.array([1, 1, 0, 1], np.float32),
            "tiny": synthetic.random_lights(8, 100, radius=(1e-4, 1e-3)),
            "many": synthetic.random_lights(9, 5000, radius=(1e-3, 3e-2))}
    for name, lights in sets.items():
        e_scene.set_lights(lights)
        assert e_scene.light_grid_info()[3] >= len(lights), name
        count = 12000
        for extent in (6.0, 300.0, 5000.0):
            origins = rng.uniform(-extent, extent, (count, 3)).astype(np.float32)
            aimed = lights[rng.integers(0, len(lights), count)]
            # Towards a point on or just off the silhouette of a light, or anywhere
            offset = unit(rng.normal(size=(count, 3))) * aimed[:, 3:4] * rng.choice([0.3, 0.999, 1.0, 1.001], (count, 1))
            directions = np.where(rng.uniform(size=(count, 1)) < 0.6, unit(aimed[:, :3] + offset - origins), unit(rng.normal(size=(count, 3))))
            kind = rng.integers(0, 6, count)
            directions[kind == 0, 0] = 0
            directions[kind == 1, 1:] = 0
            directions = (directions * rng.choice([1.0, 1.0, 0.01, 7.0], (count, 1))).astype(np.float32)
            mismatches, overflows, hits = e_scene.light_grid_mismatches(origins, directions)
            assert mismatches == 0, (name, extent, mismatches)
            normals = unit(rng.normal(size=(count, 3)))
            others = unit(rng.normal(size=(count, 3)))
            by_grid, by_walk, by_loop = (e_scene.lights_density(origins, normals, directions, others, mode) for mode in range(3))
            assert (bits(by_grid) == bits(by_loop)).all() and (bits(by_walk) == bits(by_loop)).all(), (name, extent)
            assert (by_loop[:, 0] > 0).sum() > count // 20 or name == "tiny", (name, extent)
            assert e_scene.light_tree_reuse_mismatches(origins, normals, rng.uniform(size=(count, 2))) == 0, name
    e_scene.close()


def test_light_tree_degenerate_lights():
    """No lights, lights of radius zero and coincident lights neither crash nor yield NaN where the loop does not."""
    scene = synthetic.triangle_soup(seed=3, triangle_count=500, material_count=2)
    o_scene, e_scene = oracle.OracleScene(scene), emu.EmuScene(scene)
    spec = synthetic.look_at_spec((0.5, 7.0, 0.3), (1.3, 0.0, 0.4))
    for lights in (np.zeros((0, 4), np.float32), np.array([[1, 2, 3, 0]], np.float32), np.array([[1, 2, 3, 0.5]] * 3 + [[4, 4, 4, 0.0]], np.float32)):
        e_scene.set_lights(lights)
        constants = oracle.write_constants(spec, 24, 16, scene, lights)
        a, rays_a = o_scene.render(constants, lights, strategy=3, path_length=3, first_frame_index=1, frame_count=1, light_sampling=1)
        b, rays_b = e_scene.render(constants, strategy=4, path_length=3, first_frame_index=1, frame_count=1)
        assert rays_a == rays_b
        assert ((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all()
        assert np.isfinite(a).all()
    # Lights that are not numbers must not break the builders (their sort needs a total order); what they render
    # is as undefined as in the reference, where a NaN importance poisons the sum over all lights
    lights = np.array([[1, 2, 3, 0.5], [np.nan, 0, 0, 0.2], [0, np.inf, 0, 0.1], [4, 4, 4, np.nan], [-2, 1, 0, 0.3]] * 3, np.float32)
    e_scene.set_lights(lights)
    constants = oracle.write_constants(spec, 24, 16, scene, lights[:1])
    o_scene.render(constants, lights, strategy=3, path_length=3, first_frame_index=1, frame_count=1, light_sampling=1)
    e_scene.render(constants, strategy=4, path_length=3, first_frame_index=1, frame_count=1)


def test_light_tree_converges_to_the_loop_over_all_lights():
    """Unbiasedness: direct light on a floor below 100 emitters, many samples. The tree estimate, the reference's
    loop over all lights and BRDF sampling (which involves no light list at all) agree in the image mean and per
    pixel within their noise; a tree that dropped or double-counted lights, or whose density did not match its
    sampling, would not. The camera sees the floor only: on an emitter's surface a shading point lies inside the
    light's sphere, where the reference's sampling is degenerate and its estimate differs from BRDF sampling."""
    scene, lights = synthetic.lit_room(seed=8, light_count=100)
    o_scene = oracle.OracleScene(scene)
    spec = synthetic.look_at_spec((3.0, -4.0, 0.6), (0.6, 0.0, 2.5), sky_strength=0.0)
    constants = oracle.write_constants(spec, 24, 16, scene, lights)
    assert (o_scene.primary_hits(constants, 1)[0] == 0).all()
    spp = 1024
    images = {}
    for i, (name, strategy, light_sampling) in enumerate([("loop", 3, 0), ("tree", 3, 1), ("loop again", 3, 0), ("brdf", 2, 0)]):
        accum, _ = o_scene.render(constants, lights, strategy=strategy, path_length=2, first_frame_index=1 + i * spp, frame_count=spp, threads=8, light_sampling=light_sampling)
        images[name] = accum[..., :3] / spp
    mean = images["loop"].mean()
    assert mean > 0.05
    relative = {name: abs(image.mean() - mean) / mean for name, image in images.items()}
    rmse = {name: np.sqrt(((image - images["loop"]) ** 2).mean()) / mean for name, image in images.items()}
    print("relative differences of image means", relative, "relative RMSE", rmse)
    assert relative["tree"] < 0.015 and relative["loop again"] < 0.015 and relative["brdf"] < 0.05
    assert rmse["tree"] < 2.0 * rmse["loop again"] + 0.02


def axis_aligned_terraces(cells=24):
    """Axis-aligned squares at two heights with walls between them: boxes without thickness, many of them on the
    boundaries of BVH nodes, where a box test that is too tight loses hits."""
    rng = np.random.default_rng(1)
    tris = []
    for i in range(cells):
        for j in range(cells):
            x0, y0, step = i / cells, j / cells, 1.0 / cells
            z = 0.0 if (i + j) % 2 == 0 else 0.5
            a, b, c, d = (x0, y0, z), (x0 + step, y0, z), (x0 + step, y0 + step, z), (x0, y0 + step, z)
            tris += [[a, b, c], [a, c, d], [(x0, y0, 0.0), (x0, y0 + step, 0.0), (x0, y0 + step, 0.5)]]
    verts = np.array(tris, np.float64)
    positions, factor, summand = synthetic.quantize_positions(verts.reshape(-1, 3))
    count = len(tris)
    normals_uvs = np.zeros((3 * count, 4), np.uint16)
    normals_uvs[:, :2] = 32768
    textures = [synthetic.random_texture(rng, 4, 132), synthetic.random_texture(rng, 4, 131), synthetic.random_texture(rng, 4, 141)]
    return oracle.formats.Scene(["m"], count, factor, summand, positions, normals_uvs, np.zeros(count, np.uint8), textures)


def far_rays(rng, count, distance, cells=24):
    """Rays towards points of the terraces (half of them on cell edges, a third along the axes) from `distance` away."""
    targets = rng.uniform(0, 1, (count, 3))
    targets[:, 2] = rng.choice([0.0, 0.5, 0.25], count)
    targets[::2, 0] = np.round(targets[::2, 0] * cells) / cells
    directions = rng.normal(size=(count, 3))
    directions[::3] = np.round(directions[::3] / np.abs(directions[::3]).max(axis=1, keepdims=True))
    directions /= np.linalg.norm(directions, axis=1, keepdims=True)
    origins = targets - directions * distance * rng.uniform(0.5, 2.0, (count, 1))
    return origins.astype(np.float32), directions.astype(np.float32)


@pytest.mark.parametrize("distance", [1.0, 100.0, 10000.0])
def test_traversal_against_a_loop_over_all_triangles(distance):
    """The closest hit must not depend on the BVH: traversal of the 8-wide BVH (kernel body) and of the oracle's
    binary BVH against a loop over all triangles with the same triangle test, for rays that start up to 10 000
    scene extents away. There the rounding of 'box minus ray origin' exceeds any padding fixed at build time; the
    box tests give way in proportion to the distance instead."""
    scene = axis_aligned_terraces()
    e_scene, o_scene = emu.EmuScene(scene), oracle.OracleScene(scene)
    origins, directions = far_rays(np.random.default_rng(int(distance)), 15000, distance)
    expected = e_scene.trace_rays_brute_force(origins, directions)
    assert (expected[0] >= 0).mean() > 0.6
    for got in (e_scene.trace_rays(origins, directions), o_scene.trace_rays(origins, directions)):
        assert (got[0] == expected[0]).all() and (bits(got[1]) == bits(expected[1])).all() and (got[2] == expected[2]).all()

