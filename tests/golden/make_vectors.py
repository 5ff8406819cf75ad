"""Writes regression vectors of the CPU oracle: small renders of the Cornell box with each sampling strategy and with
the light tree, as float32 arrays. They come from this repository's own oracle, not from the reference (which
cannot run here): they pin the oracle against drift, so that a change that alters both the oracle and the HIP
library in the same way does not pass unnoticed. Regenerate only when a change of results is intended:
python tests/golden/make_vectors.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from oracle import formats  # noqa: E402

# name, sampling strategy, light sampling, extra lights besides the one of the scene file
EXTRA_LIGHTS = [[0.1, -0.2, 0.3, 0.03], [0.45, -0.4, 0.1, 0.05], [0.3, -0.1, 0.5, 0.02], [0.2, -0.5, 0.45, 0.04], [0.5, -0.3, 0.52, 0.01]]
CASES = [("spherical", 0, 0, 0), ("psa", 1, 0, 0), ("brdf", 2, 0, 0), ("nee", 3, 0, 0), ("nee_6_lights", 3, 0, 5), ("nee_light_tree_6_lights", 3, 1, 5)]
WIDTH, HEIGHT, FIRST_FRAME, FRAMES, PATH_LENGTH = 32, 24, 2696, 2, 4


def render_cases():
    data = os.path.join(HERE, "data")
    scene = formats.load_scene(os.path.join(data, "cornell_box.vks"), os.path.join(data, "cornell_box_textures"))
    lights = formats.load_lights(os.path.join(data, "cornell_box.lights"))
    spec = formats.load_rt_save(os.path.join(data, "saves", "cornell_box", "default.rt_save"))
    constants = oracle.write_constants(spec, WIDTH, HEIGHT, scene, lights)
    o_scene = oracle.OracleScene(scene)
    for name, strategy, light_sampling, extra in CASES:
        case_lights = np.concatenate([lights, np.array(EXTRA_LIGHTS[:extra], np.float32).reshape(-1, 4)])
        image, rays = o_scene.render(constants, case_lights, strategy=strategy, path_length=PATH_LENGTH, first_frame_index=FIRST_FRAME,
                                     frame_count=FRAMES, threads=4, light_sampling=light_sampling)
        yield name, image, rays


def main():
    rays = {}
    for name, image, count in render_cases():
        np.save(os.path.join(HERE, "vectors", "cornell_%s.npy" % name), image)
        rays[name] = count
    with open(os.path.join(HERE, "vectors", "ray_counts.txt"), "w") as file:
        for name, count in rays.items():
            file.write("%s %d\n" % (name, count))


if __name__ == "__main__":
    main()
