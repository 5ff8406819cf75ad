"""Randomised parity run on a GPU box: small random scenes, cameras, light sets, sampling strategies and path lengths,
the library against the CPU oracle, bit for bit. Not part of the test suites (it is open-ended); a failure prints the
case's parameters so that it can be turned into a test.
usage: python scripts/gpu_fuzz_parity.py [--seconds 120] [--seed 1]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import oracle  # noqa: E402
from path_tracer_amd import hip_api  # noqa: E402
from tests import synthetic  # noqa: E402
from tests.test_gpu_parity import gpu_scene, same, to_gpu_constants  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--seconds", type=float, default=120.0)
    parser.add_argument("--seed", type=int, default=1)
    args = parser.parse_args()
    rng = np.random.default_rng(args.seed)
    begin, cases, failures, lit_cases, rays = time.time(), 0, 0, 0, 0
    while time.time() - begin < args.seconds:
        case = dict(scene_seed=int(rng.integers(1, 10**6)), triangles=int(rng.choice([3, 60, 900, 7000])), materials=int(rng.integers(1, 6)),
                    texture_size=int(rng.choice([4, 16, 64])), light_seed=int(rng.integers(1, 10**6)),
                    lights=int(rng.choice([0, 1, 5, 32, 33, 200, 1500])), light_extent=float(rng.choice([2.0, 10.0, 60.0])),
                    radius=float(rng.choice([0.01, 0.3, 2.0])), strategy=int(rng.choice([0, 1, 2, 3, 3, 3, 3, 3])), path_length=int(rng.choice([0, 1, 2, 3, 4, 4, 5])),
                    tree=bool(rng.integers(0, 2)), width=int(rng.choice([17, 64, 96])), height=int(rng.choice([9, 40, 64])),
                    first=int(rng.integers(0, 70000)), frames=int(rng.integers(1, 4)), camera=int(rng.integers(0, 4)),
                    eye=[float(v) for v in (np.array([0.5, 7.0, 0.3]) + rng.normal(scale=1.5, size=3))], layout=str(rng.choice(["random"] * 4 + list(synthetic.LIGHT_LAYOUTS))))
        scene = synthetic.triangle_soup(seed=case["scene_seed"], triangle_count=case["triangles"], material_count=case["materials"], texture_size=case["texture_size"])
        if case["layout"] == "random":
            lights = synthetic.random_lights(case["light_seed"], case["lights"], extent=case["light_extent"], radius=(0.3 * case["radius"], case["radius"]))
        else:
            lights = synthetic.light_layout(case["layout"])
        spec = synthetic.look_at_spec(tuple(case["eye"]), (1.3, 0.0, 0.4), camera_type=case["camera"])
        constants = oracle.write_constants(spec, case["width"], case["height"], scene, lights)
        g_scene = gpu_scene(scene, lights)
        g_scene.set_light_sampling(hip_api.LIGHT_SAMPLING_TREE if case["tree"] else hip_api.LIGHT_SAMPLING_EXACT)
        g_scene.create_render_targets(case["width"], case["height"])
        g_scene.set_render_settings(case["strategy"], case["path_length"], scene.emission_material_index)
        g_scene.reset_statistics()
        g_scene.render_samples(to_gpu_constants(constants), case["first"], case["frames"], 1, accumulate=False)
        got, _ = g_scene.read_accumulation()
        stats = g_scene.statistics()
        expected, expected_rays = oracle.OracleScene(scene).render(constants, lights, strategy=case["strategy"], path_length=case["path_length"],
                                                                   first_frame_index=case["first"], frame_count=case["frames"], threads=16,
                                                                   light_sampling=1 if case["tree"] else 0)
        ok = same(got, expected) and stats.extension_rays + stats.shadow_rays == expected_rays
        # With the light tree no vector that is not a direction reaches the BRDF: radiance is not negative
        # (the loops over all lights follow the reference, which does go negative inside the sphere of a light)
        if case["tree"] and case["strategy"] == 3 and not (np.isnan(got[..., :3]) | (got[..., :3] >= 0.0)).all():
            ok = False
            print("NEGATIVE RADIANCE", float(np.nanmin(got[..., :3])), flush=True)
        g_scene.close()
        cases += 1
        lit_cases += stats.shadow_rays > 0
        rays += stats.extension_rays + stats.shadow_rays
        if not ok:
            failures += 1
            print("MISMATCH", case, flush=True)
        elif cases % 20 == 0:
            print("%d cases, %.0f s" % (cases, time.time() - begin), flush=True)
    print("%d cases (%d with shadow rays, %.1f M rays in all), %d mismatches" % (cases, lit_cases, rays / 1e6, failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
