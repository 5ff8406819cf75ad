"""Parity on generated scenes in the large-scene regime of the kernels: BVH deeper than the LDS-resident prefix,
about a hundred materials, the reference's 28 Bistro lights, hits far from the origin."""
import os

import numpy as np
import pytest

import oracle
from oracle import formats
from path_tracer_amd import host_api
from path_tracer_amd.tools import procedural
from tests.paths import DATA

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("kind,name,textures,save,triangles", [
    ("city", "Bistro_outside", "Bistro_textures", "bistro/outside.rt_save", 400000),
    ("interior", "Arcade", "Arcade_textures", "Arcade/default.rt_save", 150000)])
def test_generated_scene_exact(tmp_path_factory, kind, name, textures, save, triangles):
    root = str(tmp_path_factory.mktemp(kind))
    options = {"triangles": triangles, "texture_size": 64}
    if kind == "city":
        options["lights_file"] = os.path.join(DATA, "Bistro_outside.lights")
        options["save_file"] = os.path.join(DATA, "saves", "bistro", "outside.rt_save")
    procedural.ensure_scene(kind, root, **options)
    data = os.path.join(root, "data")
    scene = formats.load_scene(os.path.join(data, name + ".vks"), os.path.join(data, textures))
    lights = formats.load_lights(os.path.join(data, name + ".lights"))
    spec = formats.load_rt_save(os.path.join(data, "saves", save))
    width, height, frames = 256, 144, 2
    renderer = host_api.Renderer(host_api.quickload(os.path.join(data, "saves", save)), root, device=0, width=width, height=height)
    info = renderer.bvh_info()
    assert info.triangle_count == scene.triangle_count and info.node_count * info.node_bytes > 112 * 1024
    expected_constants = oracle.write_constants(spec, width, height, scene, lights)
    assert bytes(renderer.constants()) == bytes(expected_constants)
    renderer.render_frames(frames)
    got, count = renderer.read_accumulation()
    stats = renderer.statistics()
    first = int(spec["frame_index"][0]) + 1
    expected, rays = oracle.OracleScene(scene).render(expected_constants, lights, first_frame_index=first, frame_count=frames, threads=16)
    assert count == frames and stats.extension_rays + stats.shadow_rays == rays
    assert (bits(got) == bits(expected)).all()
    assert np.isfinite(got).all() and got[..., :3].mean() > 0.01
    renderer.close()
