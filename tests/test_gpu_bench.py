"""bench.py on the one GPU of a test box: its one-process-many-GPUs code path (pt_create_context,
pt_context_render_samples, pt_context_reduce) driven with a device list of one, checked against plain
pt_render_samples() calls, and its refusal to measure fewer GPUs than it was asked for."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import bench
from path_tracer_amd import hip_api, host_api
from tests.paths import DATA, DATA_ROOT, ROOT
from tests.test_gpu_parity import bits

pytestmark = pytest.mark.gpu


def test_context_code_path_of_the_benchmark(monkeypatch):
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(name, raising=False)
    width, height, spp, steps, warmup = 192, 108, 4, 3, 1
    args = bench.parse_args(["--gpus", "1", "--force-context", "--workload", "cornell", "--width", str(width), "--height", str(height), "--spp-per-step", str(spp),
                             "--steps", str(steps), "--warmup", str(warmup), "--strong-frames", "8", "--strong-steps", "2", "--no-cpu-baseline"])
    image = []
    result = bench.run(args, image_out=image)
    assert result["n_gpus"] == 1 and result["scaling"] == "weak" and "pt_create_context" in result["config"]["parallelism"]
    assert result["steps"] == steps and result["config"]["spp_total"] == steps * spp
    # Ray conservation: what the GPUs report adds up to what the line's rates were computed from
    per_gpu = result["per_gpu"]
    assert len(per_gpu) == 1 and per_gpu[0]["device"] == 0 and per_gpu[0]["paths"] == width * height * spp * steps
    rays = per_gpu[0]["extension_rays"] + per_gpu[0]["shadow_rays"]
    assert abs(rays / per_gpu[0]["paths"] - result["rays_per_path"]) < 1e-12
    seconds = result["ms_per_step"] * 1e-3 * steps
    assert abs(rays / seconds * 1e-6 - result["value"]) < 1e-6 * result["value"]
    for leg, frames in (("weak", spp), ("strong", 8)):
        assert result[leg]["frames_per_step"] == frames and result[leg]["value"] > 0 and result[leg]["all_reduce_ms"] >= 0
    assert result["shadow_skipping"]["state"] in ("on", "off") and result["shadow_skipping"]["decided_on_rays"] >= 4096
    # The image of the timed leg equals plain pt_render_samples() calls for the same frames
    got, count = image
    first = result["weak"]["first_frame_index"]
    spec = host_api.quickload(os.path.join(DATA, "saves", "cornell_box", "default.rt_save"))
    assert first == spec.frame_index + 1 + (1 + warmup) * spp
    single = host_api.Renderer(spec, DATA_ROOT, device=0, width=width, height=height)
    constants = single.constants()
    assert hip_api.lib().pt_render_samples(single.device_scene, C.byref(constants), first, steps * spp, 1, 0) == 0
    expected, expected_count = single.read_accumulation()
    single.close()
    assert count == expected_count == steps * spp
    assert (bits(got) == bits(expected)).all() and (got[..., 3] == 1.0).all()


def test_benchmark_refuses_more_gpus_than_there_are():
    visible = hip_api.lib().pt_get_device_count()
    assert visible >= 1
    environment = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    result = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(visible + 1), "--no-cpu-baseline", "--workload", "cornell"],
                            capture_output=True, text=True, timeout=300, env=environment)
    assert result.returncode != 0 and result.stdout.strip() == ""
    assert "only %d GPU" % visible in result.stderr and "refusing" in result.stderr


def test_accumulating_into_an_empty_target_takes_the_discard_pipeline():
    """pt_render_samples(accumulate = 1) on a target that holds no frame behaves like accumulate = 0: the reference
    binds the discard pipeline whenever accum_frame_count is zero (src/main.c:1884-1887), so alpha is 1."""
    spec = host_api.quickload(os.path.join(DATA, "saves", "cornell_box", "default.rt_save"))
    renderer = host_api.Renderer(spec, DATA_ROOT, device=0, width=64, height=48)
    constants = renderer.constants()
    lib = hip_api.lib()
    assert lib.pt_render_samples(renderer.device_scene, C.byref(constants), 100, 3, 1, 1) == 0
    fresh, count = renderer.read_accumulation()
    assert count == 3 and (fresh[..., 3] == 1.0).all()
    assert lib.pt_render_samples(renderer.device_scene, C.byref(constants), 100, 3, 1, 0) == 0
    restarted, _ = renderer.read_accumulation()
    assert (bits(fresh) == bits(restarted)).all()
    # and a target emptied by pt_create_render_targets() or pt_set_sample_count(0) likewise
    assert lib.pt_set_sample_count(renderer.device_scene, 0) == 0
    assert lib.pt_render_samples(renderer.device_scene, C.byref(constants), 100, 3, 1, 1) == 0
    again, count = renderer.read_accumulation()
    assert count == 3 and (bits(again) == bits(fresh)).all()
    renderer.close()


@pytest.mark.parametrize("light_sampling", [hip_api.LIGHT_SAMPLING_EXACT, hip_api.LIGHT_SAMPLING_TREE])
def test_fast_shading_math_stays_within_its_tolerance(light_sampling):
    """pt_set_shading_math(PT_SHADING_MATH_FAST): the shading kernel with fused multiply-adds, approximate division
    and square root, and bare reciprocals in the loops over all lights (or, with the light tree, in its descent and
    its probabilities). Primary hits are the same bits (they do not
    depend on it); radiance at equal seeds differs from the exact mode in the last bits nearly everywhere, and by more
    where such a difference moves a sample across the boundary between two lights or two triangles: relative RMSE
    below 2 % at 32 samples per pixel, below 1 % at 512, image means within 0.1 %."""
    from tests import synthetic
    import oracle
    from tests.test_gpu_parity import gpu_scene, to_gpu_constants
    scene, lights = synthetic.lit_room(seed=5, light_count=28)
    width, height = 256, 144
    constants = to_gpu_constants(oracle.write_constants(synthetic.look_at_spec((4.0, -5.0, 2.5), (1.3, 0.0, 0.7), emission_strength=30.0), width, height, scene, lights))
    library = gpu_scene(scene, lights)
    library.create_render_targets(width, height)
    library.set_render_settings(3, 4, scene.emission_material_index)
    library.set_light_sampling(light_sampling)
    images = {}
    for spp in (32, 512):
        for mode in (hip_api.SHADING_MATH_EXACT, hip_api.SHADING_MATH_FAST):
            library.set_shading_math(mode)
            library.render_samples(constants, 900, spp, 1, accumulate=False)
            image, count = library.read_accumulation()
            images[spp, mode] = image[..., :3].astype(np.float64) / count
            hits = library.trace_primary_rays(constants, 900)
            if mode == hip_api.SHADING_MATH_EXACT:
                exact_hits = hits
            else:
                assert all((a == b).all() for a, b in zip(hits, exact_hits))
    errors = {}
    for spp in (32, 512):
        exact, fast = images[spp, hip_api.SHADING_MATH_EXACT], images[spp, hip_api.SHADING_MATH_FAST]
        assert np.isfinite(fast).all() and exact.mean() > 0.01
        errors[spp] = np.sqrt(((fast - exact) ** 2).mean()) / exact.mean()
        assert abs(fast.mean() - exact.mean()) < 1e-3 * exact.mean(), (spp, fast.mean(), exact.mean())
    print("relative RMSE of fast against exact shading math:", errors)
    assert errors[32] < 0.02 and errors[512] < 0.01, errors
    library.set_shading_math(hip_api.SHADING_MATH_EXACT)
    library.close()


def test_benchmark_code_path_for_two_gpus_on_one(monkeypatch):
    """The benchmark's multi-GPU path with two replicas on the one GPU of the box (PT_CONTEXT_LOOPBACK=1): both
    replicas render their slices, the line adds them up, and the image is the single scene's within the rounding of a
    different order of additions."""
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(name, raising=False)
    monkeypatch.setenv("PT_CONTEXT_LOOPBACK", "1")
    width, height, spp, steps, warmup = 128, 72, 3, 2, 1
    args = bench.parse_args(["--gpus", "2", "--devices", "0,0", "--workload", "cornell", "--width", str(width), "--height", str(height), "--spp-per-step", str(spp),
                             "--steps", str(steps), "--warmup", str(warmup), "--strong-frames", "5", "--strong-steps", "2", "--no-cpu-baseline"])
    image = []
    result = bench.run(args, image_out=image)
    assert result["n_gpus"] == 2 and result["config"]["spp_total"] == steps * spp * 2
    per_gpu = result["per_gpu"]
    assert len(per_gpu) == 2 and all(g["paths"] == width * height * spp * steps for g in per_gpu)
    rays = sum(g["extension_rays"] + g["shadow_rays"] for g in per_gpu)
    assert abs(rays / (2 * per_gpu[0]["paths"]) - result["rays_per_path"]) < 1e-12
    assert result["weak"]["frames_per_step"] == 2 * spp and result["weak"]["frames_per_gpu_and_step"] == spp
    assert result["strong"]["frames_per_step"] == 5 and result["strong"]["frames_per_gpu_and_step"] == 2.5
    got, count = image
    first = result["weak"]["first_frame_index"]
    spec = host_api.quickload(os.path.join(DATA, "saves", "cornell_box", "default.rt_save"))
    assert first == spec.frame_index + 1 + (1 + warmup) * spp * 2
    single = host_api.Renderer(spec, DATA_ROOT, device=0, width=width, height=height)
    constants = single.constants()
    assert hip_api.lib().pt_render_samples(single.device_scene, C.byref(constants), first, steps * spp * 2, 1, 0) == 0
    expected, expected_count = single.read_accumulation()
    single.close()
    assert count == expected_count == steps * spp * 2 and (got[..., 3] == 1.0).all()
    np.testing.assert_allclose(got[..., :3], expected[..., :3], rtol=1e-5, atol=1e-6)


def test_benchmark_under_a_launcher_with_one_rank():
    """The one-process-per-GPU code path as torch.distributed.run starts it (WORLD_SIZE, RANK, LOCAL_RANK set): process
    group over RCCL, all-reduce inside the timed region. In a fresh process, because torch has to be loaded before the
    HIP library there."""
    import json
    environment = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    result = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collectives", "--workload", "cornell", "--width", "160", "--height", "90",
                             "--spp-per-step", "4", "--steps", "2", "--warmup", "1", "--strong-frames", "8", "--strong-steps", "1", "--no-cpu-baseline"],
                            capture_output=True, text=True, timeout=600, env=environment)
    assert result.returncode == 0, result.stderr[-2000:]
    line = json.loads(result.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "all-reduce" in line["config"]["parallelism"]
    assert line["weak"]["frames_per_step"] == 4 and line["strong"]["frames_per_step"] == 8 and line["all_reduce_ms"] > 0
    # a launcher whose world size differs from --gpus is refused
    result = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cornell", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=environment)
    assert result.returncode != 0 and "must agree" in result.stderr
