"""The device side of the multi-GPU path on the one GPU a test box has: the accumulation target, allocated by the
library and not by torch, is handed to RCCL in place."""
import os
import socket

import numpy as np
import pytest

from path_tracer_amd import host_api, multi_gpu
from tests.paths import DATA, DATA_ROOT

pytestmark = pytest.mark.gpu


def test_rccl_all_reduce_in_place_on_the_library_buffer():
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        spec = host_api.quickload(os.path.join(DATA, "saves", "cornell_box", "default.rt_save"))
        renderer = host_api.Renderer(spec, DATA_ROOT, device=0, width=128, height=72)
        first, count, stride = multi_gpu.frame_slice(0, 1, spec.frame_index + 1, 3)
        assert (first, count, stride) == (2696, 3, 1)
        renderer.render_frames(3)
        before, samples = renderer.read_accumulation()
        tensor = multi_gpu.accumulation_tensor(renderer)
        assert tensor.data_ptr() == renderer.accumulation_device_pointer() and tuple(tensor.shape) == (72, 128, 4)
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)  # one rank: the sum is the buffer itself
        torch.cuda.synchronize()
        after, _ = renderer.read_accumulation()
        assert samples == 3 and (after.view(np.uint32) == before.view(np.uint32)).all()
        # The tensor aliases the library's memory: a write through torch is what the library reads back
        tensor[..., :3] *= 2.0
        torch.cuda.synchronize()
        doubled, _ = renderer.read_accumulation()
        assert (doubled[..., :3] == 2.0 * before[..., :3]).all() and (doubled[..., 3] == 1.0).all()
        multi_gpu.all_reduce_accumulation(renderer, total_sample_count=6)
        assert renderer.read_accumulation()[1] == 6
        renderer.close()
    finally:
        dist.destroy_process_group()
