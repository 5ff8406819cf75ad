"""The N > 1 path rehearsed on CPU: two processes over the gloo backend each render their sample slice (with the
host-executed kernel bodies standing in for a GPU), sum their accumulation targets with one all-reduce and must
reproduce the single-process image up to float summation order."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from tests.paths import DATA, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import oracle
    from oracle import formats
    from path_tracer_amd import multi_gpu
    from tests import emu
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    scene = formats.load_scene(os.path.join(DATA, "cornell_box.vks"), os.path.join(DATA, "cornell_box_textures"))
    lights = formats.load_lights(os.path.join(DATA, "cornell_box.lights"))
    spec = formats.load_rt_save(os.path.join(DATA, "saves", "cornell_box", "default.rt_save"))
    constants = oracle.write_constants(spec, 40, 32, scene, lights)
    total = 7  # odd on purpose: ranks get 4 and 3 frames
    first, count, stride = multi_gpu.frame_slice(rank, world_size, 2696, total)
    renderer = emu.EmuScene(scene)
    renderer.set_lights(lights)
    image, rays = renderer.render(constants, first_frame_index=first, frame_count=count, frame_stride=stride)
    reduced = multi_gpu.all_reduce_host(image)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), reduced)
    np.save(os.path.join(out_dir, "count%d.npy" % rank), np.array([count, rays]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_slices_sum_to_the_full_image(tmp_path, cornell, cornell_lights, cornell_spec, oracle_cornell):
    import oracle
    world_size = 2
    mp.spawn(_worker, args=(world_size, _free_port(), str(tmp_path)), nprocs=world_size, join=True)
    constants = oracle.write_constants(cornell_spec, 40, 32, cornell, cornell_lights)
    full, rays = oracle_cornell.render(constants, cornell_lights, first_frame_index=2696, frame_count=7, threads=4)
    images = [np.load(tmp_path / ("rank%d.npy" % r)) for r in range(world_size)]
    counts = [np.load(tmp_path / ("count%d.npy" % r)) for r in range(world_size)]
    assert (images[0] == images[1]).all()
    assert sum(int(c[0]) for c in counts) == 7 and sum(int(c[1]) for c in counts) == rays
    # Same samples, different order of float additions: tolerance 1e-5 relative to the pixel sum
    np.testing.assert_allclose(images[0][..., :3], full[..., :3], rtol=1e-5, atol=1e-6)
    assert (images[0][..., 3] == 1).all()
