"""Several GPUs through the C ABI (pt_create_context, pt_context_render_samples, pt_context_reduce) and through
the C host, on the one GPU a test box has: a context of one GPU takes the multi-GPU code path and must produce the
bits of the single-scene path, both as it is (nothing to sum, RCCL not loaded) and with PT_CONTEXT_FORCE_RCCL=1,
which sends the reduction through RCCL (communicator, grouped ncclReduce, alpha reset) as for several GPUs. The sharing of frames among GPUs itself
(frame_stride) is covered by test_gpu_parity.py::test_sample_slices_1080p and tests/test_multi_gpu_cpu.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch  # noqa: F401  loaded before the HIP library so that both share one HIP and RCCL runtime

import oracle
from path_tracer_amd import hip_api, host_api
from tests import synthetic
from tests.paths import DATA, DATA_ROOT
from tests.test_gpu_parity import bits, gpu_scene, to_gpu_constants

pytestmark = pytest.mark.gpu


def create_context(scene, devices):
    lib = hip_api.lib()
    textures = [(t.width, t.height, t.format, [np.ascontiguousarray(t.payload[int(t.mip_offsets[m]):int(t.mip_offsets[m]) + t.mip_extents[m][2]]) for m in range(t.mip_count)])
                for t in scene.textures]
    header = hip_api.SceneHeader()
    header.material_count = len(scene.material_names)
    header.triangle_count = scene.triangle_count
    header.dequantization_factor[:] = [float(v) for v in scene.dequantization_factor]
    header.dequantization_summand[:] = [float(v) for v in scene.dequantization_summand]
    descs = (hip_api.TextureDesc * max(1, len(textures)))()
    keep = []
    for i, (width, height, fmt, mips) in enumerate(textures):
        data = (C.c_void_p * len(mips))(*[m.ctypes.data for m in mips])
        sizes = (C.c_uint64 * len(mips))(*[m.nbytes for m in mips])
        keep += [mips, data, sizes]
        descs[i] = hip_api.TextureDesc(width, height, fmt, len(mips), data, sizes)
    positions = np.ascontiguousarray(scene.positions, np.uint32)
    normals_uvs = np.ascontiguousarray(scene.normals_uvs, np.uint16)
    materials = np.ascontiguousarray(scene.materials, np.uint8)
    context = C.c_void_p()
    ids = (C.c_int * len(devices))(*devices)
    code = lib.pt_create_context(C.byref(context), ids, len(devices), C.byref(header), positions.ctypes.data_as(C.c_void_p),
                                 normals_uvs.ctypes.data_as(C.c_void_p), materials.ctypes.data_as(C.c_void_p), descs, len(textures))
    return code, context


@pytest.mark.parametrize("force_rccl", ["0", "1"])
def test_context_of_one_gpu_renders_the_bits_of_a_scene(force_rccl, monkeypatch):
    monkeypatch.setenv("PT_CONTEXT_FORCE_RCCL", force_rccl)
    lib = hip_api.lib()
    scene = synthetic.triangle_soup(seed=31, triangle_count=6000, material_count=3, texture_size=32)
    lights = synthetic.random_lights(5, 7)
    width, height = 160, 90
    constants = to_gpu_constants(oracle.write_constants(synthetic.look_at_spec((0, 9, 0), (np.pi / 2, 0, 0)), width, height, scene, lights))
    single = gpu_scene(scene, lights)
    single.set_render_settings(3, 4, scene.emission_material_index)
    single.create_render_targets(width, height)
    code, context = create_context(scene, [0])
    assert code == 0 and context.value
    try:
        assert lib.pt_context_device_count(context) == 1
        replica = C.c_void_p(lib.pt_context_scene(context, 0))
        assert replica.value and lib.pt_context_scene(context, 1) is None
        light_array = np.ascontiguousarray(lights, np.float32)
        assert lib.pt_context_set_lights(context, light_array.ctypes.data_as(C.c_void_p), len(light_array)) == 0
        assert lib.pt_context_set_render_settings(context, 3, 4, scene.emission_material_index) == 0
        assert lib.pt_context_set_light_sampling(context, hip_api.LIGHT_SAMPLING_EXACT) == 0
        assert lib.pt_context_create_render_targets(context, width, height) == 0

        def read_context():
            rgba = np.zeros((height, width, 4), np.float32)
            count = C.c_uint32()
            assert lib.pt_read_accumulation(replica, rgba.ctypes.data_as(C.c_void_p), C.byref(count)) == 0
            return rgba, count.value

        # A fresh accumulation of five frames, summed, then three more frames and a second sum
        assert lib.pt_context_render_samples(context, C.byref(constants), 40, 5, 0) == 0
        assert lib.pt_context_reduce(context) == 0
        assert lib.pt_context_last_reduce_ms(context) > 0.0
        single.render_samples(constants, 40, 5, accumulate=False)
        expected, expected_count = single.read_accumulation()
        got, count = read_context()
        assert count == expected_count == 5
        assert (bits(got) == bits(expected)).all() and (got[..., 3] == 1.0).all()
        assert lib.pt_context_render_samples(context, C.byref(constants), 45, 3, 1) == 0
        assert lib.pt_context_reduce(context) == 0
        single.render_samples(constants, 45, 3, accumulate=True)
        expected, expected_count = single.read_accumulation()
        got, count = read_context()
        assert count == expected_count == 8 and (bits(got) == bits(expected)).all()
        # Nothing to render is fine, too
        assert lib.pt_context_render_samples(context, C.byref(constants), 48, 0, 1) == 0
        assert lib.pt_context_synchronize(context) == 0
    finally:
        lib.pt_destroy_context(context)
        single.close()


def test_context_refuses_bad_device_lists():
    scene = synthetic.triangle_soup(seed=32, triangle_count=50, material_count=2, texture_size=8)
    for devices in ([0, 0], [], [4096]):
        code, context = create_context(scene, devices)
        assert code != 0 and not context.value


def test_host_on_a_device_list_matches_the_single_device_host(tmp_path):
    spec = host_api.quickload(os.path.join(DATA, "saves", "cornell_box", "default.rt_save"))
    one = host_api.Renderer(spec, DATA_ROOT, device=0, width=200, height=120)
    many = host_api.Renderer(spec, DATA_ROOT, width=200, height=120, devices=[0])
    assert many.device_context is not None and one.device_context is None
    for renderer in (one, many):
        renderer.render_frames(3)
        renderer.render_frames(2)
    a, count_a = one.read_accumulation()
    b, count_b = many.read_accumulation()
    assert count_a == count_b == 5 and (bits(a) == bits(b)).all()
    one.save_screenshot(str(tmp_path / "one.png"))
    many.save_screenshot(str(tmp_path / "many.png"))
    assert (tmp_path / "one.png").read_bytes() == (tmp_path / "many.png").read_bytes()
    one.close()
    many.close()


def test_command_line_driver_on_a_device_list(tmp_path):
    binary = os.path.join(os.path.dirname(host_api.LIB_PATH), "pt_render")
    outputs = []
    for name, flags in (("single.hdr", ["-device", "0"]), ("list.hdr", ["-devices", "0"])):
        out = tmp_path / name
        result = subprocess.run([binary, "-data", DATA_ROOT, "-scene", "1", "-w96", "-h64", "-spp", "6", "-o", str(out)] + flags, capture_output=True, text=True, timeout=120)
        assert result.returncode == 0, result.stdout + result.stderr
        outputs.append(out.read_bytes())
        if name == "list.hdr":
            assert "on 1 GPU " in result.stdout and "Summing the render targets over xGMI" in result.stdout
    assert outputs[0] == outputs[1]


@pytest.mark.parametrize("replicas", [2, 3, 8])
def test_context_logic_for_several_gpus_on_one(replicas, monkeypatch):
    """PT_CONTEXT_LOOPBACK=1: one GPU listed several times gets as many replicas, created by one host thread each,
    each with its own share of the frames. Everything a context of several GPUs does runs, except that the sum is
    taken by a kernel instead of RCCL (which refuses a GPU listed twice): slices, the clearing of the other targets
    after a reduction, the transfer of their sample counts, fewer frames than replicas, rendering on after a reduction."""
    monkeypatch.setenv("PT_CONTEXT_LOOPBACK", "1")
    lib = hip_api.lib()
    scene = synthetic.triangle_soup(seed=33, triangle_count=3000, material_count=3, texture_size=16)
    lights = synthetic.random_lights(6, 5)
    width, height = 96, 54
    constants = to_gpu_constants(oracle.write_constants(synthetic.look_at_spec((0, 9, 0), (np.pi / 2, 0, 0)), width, height, scene, lights))
    single = gpu_scene(scene, lights)
    single.set_render_settings(3, 4, scene.emission_material_index)
    single.create_render_targets(width, height)
    code, context = create_context(scene, [0] * replicas)
    assert code == 0 and context.value
    try:
        assert lib.pt_context_device_count(context) == replicas
        light_array = np.ascontiguousarray(lights, np.float32)
        assert lib.pt_context_set_lights(context, light_array.ctypes.data_as(C.c_void_p), len(light_array)) == 0
        assert lib.pt_context_set_render_settings(context, 3, 4, scene.emission_material_index) == 0
        assert lib.pt_context_create_render_targets(context, width, height) == 0
        root = C.c_void_p(lib.pt_context_scene(context, 0))

        def read(scene_handle):
            rgba = np.zeros((height, width, 4), np.float32)
            count = C.c_uint32()
            assert lib.pt_read_accumulation(scene_handle, rgba.ctypes.data_as(C.c_void_p), C.byref(count)) == 0
            return rgba, count.value

        def paths_of(scene_handle):
            stats = hip_api.Statistics()
            assert lib.pt_get_statistics(scene_handle, C.byref(stats)) == 0
            return stats.paths

        # 11 frames, then a reduction; then 2 more (fewer than replicas for 3 and 8), another reduction; then a fresh start with 1 frame
        for first, count, accumulate, expected_total in ((60, 11, 0, 11), (71, 2, 1, 13), (90, 1, 0, 1)):
            for g in range(replicas):
                assert lib.pt_reset_statistics(C.c_void_p(lib.pt_context_scene(context, g))) == 0
            assert lib.pt_context_render_samples(context, C.byref(constants), first, count, accumulate) == 0
            assert lib.pt_context_synchronize(context) == 0
            # every replica rendered its slice and nothing else
            for g in range(replicas):
                slice_first, slice_count = C.c_uint32(), C.c_uint32()
                assert lib.pt_context_frame_slice(g, replicas, first, count, C.byref(slice_first), C.byref(slice_count)) == 0
                assert paths_of(C.c_void_p(lib.pt_context_scene(context, g))) == slice_count.value * width * height
            assert lib.pt_context_reduce(context) == 0
            single.render_samples(constants, first, count, accumulate=bool(accumulate))
            expected, expected_count = single.read_accumulation()
            got, got_count = read(root)
            assert got_count == expected_count == expected_total
            assert (got[..., 3] == 1.0).all()
            np.testing.assert_allclose(got[..., :3], expected[..., :3], rtol=1e-5, atol=1e-6)
            # the other replicas start over: nothing is counted twice by the next reduction
            for g in range(1, replicas):
                other, other_count = read(C.c_void_p(lib.pt_context_scene(context, g)))
                assert other_count == 0 and (other == 0.0).all()
        # reducing again without rendering changes nothing
        before, _ = read(root)
        assert lib.pt_context_reduce(context) == 0
        after, count = read(root)
        assert count == 1 and (bits(before) == bits(after)).all()
    finally:
        lib.pt_destroy_context(context)
        single.close()


def test_loopback_needs_its_switch_and_one_gpu(monkeypatch):
    scene = synthetic.triangle_soup(seed=32, triangle_count=50, material_count=2, texture_size=8)
    monkeypatch.delenv("PT_CONTEXT_LOOPBACK", raising=False)
    code, context = create_context(scene, [0, 0])
    assert code != 0 and not context.value
    monkeypatch.setenv("PT_CONTEXT_LOOPBACK", "1")
    code, context = create_context(scene, [0, 0, 4096])
    assert code != 0 and not context.value
