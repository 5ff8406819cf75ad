"""The checks of tests/test_estimator_pins.py against the HIP library on a GPU: densities integrate to one, samplers
follow their densities, the library's samplers and densities equal the oracle's bit for bit, and a floor under a
unit sky reflects its directional albedo with all four strategies (test hook pt_evaluate_function() and plain
renders through the C ABI)."""
import numpy as np
import pytest

import oracle
from tests import estimator_checks as ec, synthetic
from tests.test_estimator_pins import DENSITIES, LIGHTS, MATERIALS, NORMAL, OUT_DIR, SAMPLERS, decoded_material, furnace_setup
from tests.test_gpu_parity import gpu_scene, same, to_gpu_constants

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def library_scene():
    scene = gpu_scene(synthetic.triangle_soup(seed=2, triangle_count=8, material_count=2, texture_size=4), LIGHTS)
    yield scene
    scene.close()


@pytest.mark.parametrize("name,function,component,material,cos_range", DENSITIES, ids=[d[0] for d in DENSITIES])
def test_density_integrates_to_one(library_scene, name, function, component, material, cos_range):
    shading_data = ec.make_shading_data(NORMAL, OUT_DIR, **material)
    integral = ec.integrate_density(library_scene.evaluate_function, function, shading_data, NORMAL, component, cells=(768, 1536))
    assert abs(integral - 1.0) < (6e-3 if name == "lights" else 3e-3), integral


@pytest.mark.parametrize("name,sampler,density,component,material", SAMPLERS, ids=[s[0] for s in SAMPLERS])
def test_sampler_follows_its_density(library_scene, name, sampler, density, component, material):
    shading_data = ec.make_shading_data(NORMAL, OUT_DIR, **material)
    chi_square, degrees, probability, unit_share = ec.chi_square_of_sampler(library_scene.evaluate_function, sampler, density, shading_data, NORMAL, component, sample_count=300000)
    assert unit_share == 1.0 and abs(probability - 1.0) < 1e-2
    assert degrees > 20 and chi_square < ec.chi_square_limit(degrees), (chi_square, degrees)


def test_functions_equal_the_oracle_bit_for_bit(library_scene):
    rng = np.random.default_rng(9)
    randoms = rng.random((20000, 2), dtype=np.float32)
    directions = rng.normal(size=(20000, 3))
    directions = (directions / np.linalg.norm(directions, axis=1, keepdims=True)).astype(np.float32)
    for material in (dict(roughness=0.006), dict(roughness=0.3, diffuse_albedo=(0.3, 0.25, 0.2)), dict(roughness=1.0, fresnel_0=(0.9, 0.6, 0.3))):
        shading_data = ec.make_shading_data(NORMAL, OUT_DIR, **material)
        for function in range(11):
            inputs = randoms if function in (ec.SAMPLE_GGX_IN_DIR, ec.SAMPLE_HEMISPHERE_PSA, ec.SAMPLE_HEMISPHERE_SPHERICAL, ec.SAMPLE_FROSTBITE_BRDF, ec.SAMPLE_LIGHTS) else directions
            got = library_scene.evaluate_function(function, shading_data, inputs)
            expected = oracle.evaluate_function(function, shading_data, inputs, LIGHTS)
            assert same(got, expected), (function, material)


@pytest.mark.parametrize("material", MATERIALS, ids=["white", "coloured", "metal"])
def test_floor_under_a_unit_sky_reflects_its_directional_albedo(material):
    scene, no_lights, constants, towards_camera = furnace_setup(material, width=64, height=64)
    decoded = decoded_material(scene)
    data = oracle.OracleScene(scene).shading_data(constants, [0], [[0.3, 0.3]], [1], [towards_camera])[0]
    expected = ec.directional_albedo(data[3:6].astype(np.float64), towards_camera, **decoded)
    library = gpu_scene(scene, no_lights)
    library.create_render_targets(64, 64)
    spp = 1024
    for strategy in range(4):
        library.set_render_settings(strategy, 2, scene.emission_material_index)
        library.render_samples(to_gpu_constants(constants), 7, spp, 1, accumulate=False)
        image, count = library.read_accumulation()
        radiance = image[..., :3].astype(np.float64) / count
        mean = radiance.mean(axis=(0, 1))
        standard_error = radiance.reshape(-1, 3).std(axis=0) / np.sqrt(radiance.shape[0] * radiance.shape[1])
        assert (np.abs(mean - expected) < np.maximum(5.0 * standard_error, 1e-3 * expected)).all(), (strategy, mean, expected, standard_error)
    # and the bits of the oracle on a few frames
    library.set_render_settings(3, 2, scene.emission_material_index)
    library.render_samples(to_gpu_constants(constants), 7, 3, 1, accumulate=False)
    got, _ = library.read_accumulation()
    reference, _ = oracle.OracleScene(scene).render(constants, no_lights, strategy=3, path_length=2, first_frame_index=7, frame_count=3, threads=16)
    assert same(got, reference)
    library.close()


def test_a_surface_inside_the_sphere_of_a_light():
    """tests/test_estimator_pins.py::test_a_surface_inside_the_sphere_of_a_light through the library: the loops over
    all lights follow the reference into negative radiance, the light tree takes no sample of a light from inside its
    sphere and renders the bits of a scene without lights; both equal the oracle."""
    from path_tracer_amd import hip_api
    material = dict(base_color=(0.8, 0.8, 0.8), roughness=0.9, metalicity=0.0)
    scene, no_lights, constants, towards_camera = furnace_setup(material)
    light = np.array([[0.0, 0.0, -0.3, 1.0]], np.float32)
    checker = oracle.OracleScene(scene)
    spp = 64
    images = {}
    for name, lights, light_sampling in (("plain", no_lights, 0), ("loops", light, 0), ("tree", light, 1)):
        library = gpu_scene(scene, lights)
        library.create_render_targets(16, 16)
        library.set_render_settings(3, 2, scene.emission_material_index)
        library.set_light_sampling(hip_api.LIGHT_SAMPLING_TREE if light_sampling else hip_api.LIGHT_SAMPLING_EXACT)
        library.render_samples(to_gpu_constants(constants), 3, spp, 1, accumulate=False)
        images[name], _ = library.read_accumulation()
        library.close()
        reference, _ = checker.render(constants, lights, strategy=3, path_length=2, first_frame_index=3, frame_count=spp, threads=16, light_sampling=light_sampling)
        assert same(images[name], reference), name
    assert images["loops"][..., :3].min() < 0.0 and images["plain"][..., :3].min() > 0.0
    assert same(images["tree"], images["plain"])
