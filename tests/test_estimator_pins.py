"""Pins of the estimators that need no reference run: every probability density integrates to one, every sampler
is distributed like its density (chi-square over the sphere), a surface under a uniform sky reflects its
directional albedo (computed from a double-precision numpy restatement of brdfs.glsl:19-46) with each of the four
sampling strategies, and radiance under a unit sky never exceeds one and grows with the path length. Run here
against the CPU oracle and against the kernel bodies compiled for the host; tests/test_gpu_estimator_pins.py runs
the same checks against the HIP library."""
import numpy as np
import pytest

import oracle
from tests import emu, estimator_checks as ec, synthetic

NORMAL = np.array([0.3, -0.5, 0.8]) / np.linalg.norm([0.3, -0.5, 0.8])
OUT_DIR = np.array([0.7, 0.1, 0.6]) / np.linalg.norm([0.7, 0.1, 0.6])
# all of them above the horizon of NORMAL at the origin, one pair overlapping as seen from there
LIGHTS = np.array([[1.5, -2.0, 4.0, 1.2], [-2.0, -3.0, 2.0, 0.9], [0.5, -1.0, 1.2, 0.35], [0.8, -1.3, 2.4, 0.5]], np.float32)


def evaluators():
    def oracle_evaluate(function, shading_data, inputs):
        return oracle.evaluate_function(function, shading_data, inputs, LIGHTS)
    soup = synthetic.triangle_soup(seed=2, triangle_count=8, material_count=2, texture_size=4)
    emulated = emu.EmuScene(soup)
    emulated.set_lights(LIGHTS)
    return {"oracle": oracle_evaluate, "kernel bodies on the host": emulated.evaluate_function}


EVALUATORS = evaluators()
DENSITIES = [("ggx 0.15", ec.GGX_IN_DIR_DENSITY, 0, dict(roughness=0.15), (-1.0, 1.0)), ("ggx 0.5", ec.GGX_IN_DIR_DENSITY, 0, dict(roughness=0.5), (-1.0, 1.0)),
             ("ggx 1.0", ec.GGX_IN_DIR_DENSITY, 0, dict(roughness=1.0), (-1.0, 1.0)),
             ("projected solid angle", ec.HEMISPHERE_PSA_DENSITY, 0, {}, (-1.0, 1.0)), ("spherical", ec.HEMISPHERE_SPHERICAL_DENSITY, 0, {}, (-1.0, 1.0)),
             ("mixture", ec.FROSTBITE_BRDF_DENSITY, 0, dict(roughness=0.3, diffuse_albedo=(0.3, 0.25, 0.2)), (-1.0, 1.0)),
             ("lights", ec.LIGHTS_DENSITY, 1, {}, (-1.0, 1.0))]
SAMPLERS = [("ggx 0.15", ec.SAMPLE_GGX_IN_DIR, ec.GGX_IN_DIR_DENSITY, 0, dict(roughness=0.15)), ("ggx 0.6", ec.SAMPLE_GGX_IN_DIR, ec.GGX_IN_DIR_DENSITY, 0, dict(roughness=0.6)),
            ("projected solid angle", ec.SAMPLE_HEMISPHERE_PSA, ec.HEMISPHERE_PSA_DENSITY, 0, {}),
            ("spherical", ec.SAMPLE_HEMISPHERE_SPHERICAL, ec.HEMISPHERE_SPHERICAL_DENSITY, 0, {}),
            ("mixture", ec.SAMPLE_FROSTBITE_BRDF, ec.FROSTBITE_BRDF_DENSITY, 0, dict(roughness=0.3, diffuse_albedo=(0.3, 0.25, 0.2))),
            ("mixture, mostly diffuse", ec.SAMPLE_FROSTBITE_BRDF, ec.FROSTBITE_BRDF_DENSITY, 0, dict(roughness=0.5, diffuse_albedo=(0.9, 0.9, 0.9))),
            ("lights", ec.SAMPLE_LIGHTS, ec.LIGHTS_DENSITY, 1, {})]


@pytest.mark.parametrize("implementation", sorted(EVALUATORS))
@pytest.mark.parametrize("name,function,component,material,cos_range", DENSITIES, ids=[d[0] for d in DENSITIES])
def test_density_integrates_to_one(implementation, name, function, component, material, cos_range):
    """brdfs.glsl:88-121,151-153,203-208; pathtrace.frag.glsl:60-64,150-175"""
    shading_data = ec.make_shading_data(NORMAL, OUT_DIR, **material)
    integral = ec.integrate_density(EVALUATORS[implementation], function, shading_data, NORMAL, component, cells=(768, 1536))
    assert abs(integral - 1.0) < (6e-3 if name == "lights" else 3e-3), integral


@pytest.mark.parametrize("implementation", sorted(EVALUATORS))
@pytest.mark.parametrize("name,sampler,density,component,material", SAMPLERS, ids=[s[0] for s in SAMPLERS])
def test_sampler_follows_its_density(implementation, name, sampler, density, component, material):
    """brdfs.glsl:58-75,104-107,139-146,174-199; pathtrace.frag.glsl:50-55,95-143"""
    shading_data = ec.make_shading_data(NORMAL, OUT_DIR, **material)
    chi_square, degrees, probability, unit_share = ec.chi_square_of_sampler(EVALUATORS[implementation], sampler, density, shading_data, NORMAL, component, sample_count=300000)
    assert unit_share == 1.0
    assert abs(probability - 1.0) < 1e-2, probability
    assert degrees > 20 and chi_square < ec.chi_square_limit(degrees), (chi_square, degrees)


def test_lights_below_the_horizon_are_never_sampled():
    """get_spherical_light_importance() is zero for a light entirely below the horizon (pathtrace.frag.glsl:72-75)
    and sample_lights() returns the zero vector when no light is left (:142)."""
    below = LIGHTS.copy()
    below[:, :3] = -below[:, :3]
    shading_data = ec.make_shading_data(NORMAL, OUT_DIR)
    randoms = np.random.default_rng(3).random((1000, 2), dtype=np.float32)
    out = oracle.evaluate_function(ec.SAMPLE_LIGHTS, shading_data, randoms, below)
    assert (out == 0.0).all()
    mixed = np.concatenate([LIGHTS[:2], below[2:]])
    out = oracle.evaluate_function(ec.SAMPLE_LIGHTS, shading_data, randoms, mixed)
    assert (out[:, :3].astype(np.float64) @ NORMAL > 0.0).all() and (out[:, 3] > 0).all()


MATERIALS = [dict(base_color=(1.0, 1.0, 1.0), roughness=0.5, metalicity=0.0), dict(base_color=(0.8, 0.5, 0.3), roughness=0.25, metalicity=0.0),
             dict(base_color=(0.9, 0.7, 0.4), roughness=0.4, metalicity=1.0)]


def furnace_setup(material, walls=False, width=16, height=16, tilt=0.7):
    """Scene, lights, constants and the unit vector from the surface towards the eye for an orthographic view of the
    floor. The reference's orthographic camera is peculiar and reproduced as it is: its projection follows glOrtho
    (src/camera.c:139-152), so the ray origins at depth 0 (camera_utilities.glsl:10) lie halfway between the near
    and the far plane, and for a matrix whose last row is (0, 0, 0, 1) get_camera_ray_direction() (:24-44) yields
    the third row of world to view space, which points from that plane back towards the camera position. The eye
    is therefore (near + far) / 2 in front of the camera position, looking back at it."""
    scene = synthetic.furnace_scene(walls=walls, **material)
    no_lights = np.zeros((0, 4), np.float32)
    distance = 1.2 if walls else 20.0
    spec = synthetic.look_at_spec((0.0, 0.0, 0.2 if walls else 0.0), (np.pi - tilt, 0.0, 0.3), sky=(1.0, 1.0, 1.0), sky_strength=1.0, emission_strength=0.0, camera_type=1)
    spec["height"] = 1.0
    spec["near"] = 0.01
    spec["far"] = 2.0 * distance - 0.01
    constants = oracle.write_constants(spec, width, height, scene, no_lights)
    ray_dir = np.array(constants.hemispherical_camera_normal, np.float64)
    assert ray_dir[2] < -0.5
    return scene, no_lights, constants, -ray_dir


def decoded_material(scene):
    """What get_shading_data() makes of the constant textures of the surface material (shading_data.glsl:74-99)."""
    s = oracle.OracleScene(scene)
    base, specular = (s.sample_texture(3 + i, [[0.3, 0.3]])[0].astype(np.float64) for i in range(2))
    normal = s.sample_texture(5, [[0.3, 0.3]])[0].astype(np.float64)
    assert abs(normal[0] * 2 - 1) < 5e-3 and abs(normal[1] * 2 - 1) < 5e-3
    metalicity = specular[2]
    return dict(diffuse_albedo=base - metalicity * base, fresnel_0=0.02 * (1 - metalicity) + base * metalicity, roughness=max(0.006, specular[1] ** 2))


@pytest.mark.parametrize("material", MATERIALS, ids=["white", "coloured", "metal"])
def test_floor_under_a_unit_sky_reflects_its_directional_albedo(material):
    """White furnace for a convex surface, all four sampling strategies (pathtrace.frag.glsl:224-362): the radiance
    towards the camera is the integral of BRDF times cosine, which is computed here without any code of the oracle."""
    scene, no_lights, constants, towards_camera = furnace_setup(material)
    decoded = decoded_material(scene)
    # The flat normal texture decodes to 127.5 / 255 only approximately: the shading normal is tilted by what is left
    triangles, _, front = oracle.OracleScene(scene).primary_hits(constants, 1)
    assert (triangles >= 0).all() and front.all()
    data = oracle.OracleScene(scene).shading_data(constants, [0], [[0.3, 0.3]], [1], [towards_camera])[0]
    shading_normal = data[3:6].astype(np.float64)
    assert shading_normal[2] > 0.999
    expected = ec.directional_albedo(shading_normal, towards_camera, **decoded)
    assert (expected < 1.05).all() and (expected > 0.3).all()
    s = oracle.OracleScene(scene)
    spp = 1024
    for strategy in range(4):
        image, rays = s.render(constants, no_lights, strategy=strategy, path_length=2, first_frame_index=7, frame_count=spp, threads=8)
        radiance = image[..., :3].astype(np.float64) / spp
        # every pixel is an estimate of the same integral
        mean = radiance.mean(axis=(0, 1))
        standard_error = radiance.reshape(-1, 3).std(axis=0) / np.sqrt(radiance.shape[0] * radiance.shape[1])
        assert (np.abs(mean - expected) < np.maximum(5.0 * standard_error, 2e-3 * expected)).all(), (strategy, mean, expected, standard_error)
        # and the kernel bodies on the host compute the bits of the oracle
    emulated = emu.EmuScene(scene)
    emulated.set_lights(no_lights)
    for strategy in (0, 3):
        expected_image, _ = s.render(constants, no_lights, strategy=strategy, path_length=2, first_frame_index=7, frame_count=2, threads=8)
        got, _ = emulated.render(constants, strategy=strategy, path_length=2, first_frame_index=7, frame_count=2)
        assert (got.view(np.uint32) == expected_image.view(np.uint32)).all()


def test_furnace_in_a_box():
    """A sky of radiance one, no emitter, an open box in which light bounces: pathtrace.frag.glsl adds nothing but
    emission and sky, so where no direction reflects more than it receives, radiance grows with the path length and
    stays below one, for every strategy. (With a base colour of one the reference's BRDF reflects up to 2 % more
    than it receives, and 3 % more towards grazing directions at a base colour of 0.9 (Disney diffuse plus the
    specular lobe), so the box is a grey of 0.8.)"""
    material = dict(base_color=(0.8, 0.8, 0.8), roughness=0.5, metalicity=0.0)
    scene, no_lights, constants, towards_camera = furnace_setup(material, walls=True, tilt=0.5)
    decoded = decoded_material(scene)
    albedos = [ec.directional_albedo([0, 0, 1.0], np.array(out_dir) / np.linalg.norm(out_dir), cells=(256, 512), **decoded)
               for out_dir in ([0, 0, 1.0], [0.6, 0, 0.8], [0.95, 0, 0.3122], [0.999, 0, 0.0447])]
    assert (np.array(albedos) < 1.0).all() and (np.array(albedos) > 0.6).all()
    s = oracle.OracleScene(scene)
    spp = 256
    for strategy in range(4):
        means = []
        for path_length in (1, 2, 3, 5, 9):
            image, _ = s.render(constants, no_lights, strategy=strategy, path_length=path_length, first_frame_index=3, frame_count=spp, threads=8)
            radiance = image[..., :3].astype(np.float64) / spp
            assert np.isfinite(radiance).all()
            means.append(radiance.mean())
        assert means[0] == 0.0  # every camera ray hits the box, nothing in it emits
        assert all(b > a for a, b in zip(means, means[1:])), (strategy, means)
        assert 0.4 < means[-1] < 1.0, (strategy, means)


def test_a_surface_inside_the_sphere_of_a_light():
    """A floor under a sky of radiance one and a light of radius one whose centre lies 0.3 below the floor (no
    emitter: the light only steers the sampling). Seen from inside the sphere the formulas of
    pathtrace.frag.glsl:69-106 leave their domain: z_range exceeds one, z = 1 - z_range * u goes below -1, the vector
    returned points away from the centre, out of the floor, with a length of two and more, and the Fresnel factors
    of the diffuse lobe (brdfs.glsl:29-31) are negative for such a "cosine". The loops over all lights follow the
    reference there, negative radiance included (pinned here as its behaviour, not endorsed). The light tree, which
    has no counterpart in the reference, takes no sample of a light from inside its sphere, in line with its density,
    which counts a light only for rays that start outside of it: what is left is the BRDF sample with a light density
    of zero, the same bits as a render without any light; kernel body and oracle agree."""
    material = dict(base_color=(0.8, 0.8, 0.8), roughness=0.9, metalicity=0.0)
    scene, no_lights, constants, towards_camera = furnace_setup(material)
    light = np.array([[0.0, 0.0, -0.3, 1.0]], np.float32)
    lit_constants = constants  # the lights of a render are an argument of their own
    s = oracle.OracleScene(scene)
    spp = 64
    plain, plain_rays = s.render(constants, no_lights, strategy=3, path_length=2, first_frame_index=3, frame_count=spp, threads=4)
    assert plain[..., :3].min() > 0.0
    loops, _ = s.render(lit_constants, light, strategy=3, path_length=2, first_frame_index=3, frame_count=spp, threads=4)
    assert np.isfinite(loops).all() and loops[..., :3].min() < 0.0
    tree, tree_rays = s.render(lit_constants, light, strategy=3, path_length=2, first_frame_index=3, frame_count=spp, threads=4, light_sampling=1)
    assert tree_rays == plain_rays and (tree.view(np.uint32) == plain.view(np.uint32)).all()
    emulated = emu.EmuScene(scene)
    emulated.set_lights(light)
    got, got_rays = emulated.render(lit_constants, strategy=4, path_length=2, first_frame_index=3, frame_count=spp)
    assert got_rays == tree_rays and (got.view(np.uint32) == tree.view(np.uint32)).all()


def test_estimators_agree_per_channel(cornell, cornell_lights, cornell_spec, oracle_cornell):
    """The four sampling strategies estimate the same integral on the reference's own scene, light included
    (pathtrace.frag.glsl:224,258,286,315): image means per channel within their standard errors."""
    constants = oracle.write_constants(cornell_spec, 16, 16, cornell, cornell_lights)
    batches, frames = 8, 512
    results = []
    for strategy in range(4):
        batch_means = []
        for b in range(batches):
            image, _ = oracle_cornell.render(constants, cornell_lights, strategy=strategy, path_length=3, first_frame_index=1 + b * frames, frame_count=frames, threads=8)
            batch_means.append(image[..., :3].astype(np.float64).mean(axis=(0, 1)) / frames)
        batch_means = np.array(batch_means)
        results.append((batch_means.mean(axis=0), batch_means.std(axis=0, ddof=1) / np.sqrt(batches)))
    reference_mean, reference_error = results[3]
    assert (reference_error < 0.01 * reference_mean).all()
    for strategy in range(3):
        mean, error = results[strategy]
        assert (np.abs(mean - reference_mean) < 4.5 * np.sqrt(error ** 2 + reference_error ** 2)).all(), (strategy, mean, reference_mean, error)
        assert (np.abs(mean - reference_mean) < 0.06 * reference_mean).all(), (strategy, mean, reference_mean)
