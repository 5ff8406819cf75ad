"""Multi-GPU rendering: one process per GPU, independent sample slices, one sum over xGMI.

Samples are independent given (pixel, frame index) (src/shaders/pathtrace.frag.glsl:368 of the reference), so the
scene is replicated on every GPU, rank r of G renders the frame indices first + r, first + r + G, ... into its own
RGBA32F target, and one all-reduce (RCCL when the process group uses the "nccl" backend) sums the targets. The
union of the slices is exactly the sample set a single GPU would render; only the order of the float additions
differs. There is no other exchange on this path.
"""
import ctypes as C

import numpy as np


def frame_slice(rank, world_size, first_frame_index, total_sample_count):
    """(first_frame_index, sample_count, frame_stride) of one rank's slice of total_sample_count frames."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of size %d" % (rank, world_size))
    count = (total_sample_count - rank + world_size - 1) // world_size if total_sample_count > rank else 0
    return first_frame_index + rank, count, world_size


def slice_frame_indices(rank, world_size, first_frame_index, total_sample_count):
    first, count, stride = frame_slice(rank, world_size, first_frame_index, total_sample_count)
    return [first + i * stride for i in range(count)]


class DeviceBufferView:
    """Exposes a raw device allocation to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, pointer, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(pointer), False), "version": 2, "strides": None}


def accumulation_tensor(renderer):
    """The renderer's accumulation target as a torch tensor aliasing the library's device memory."""
    import torch
    pointer = renderer.accumulation_device_pointer()
    view = DeviceBufferView(pointer, (renderer.height, renderer.width, 4))
    return torch.as_tensor(view, device="cuda")


def all_reduce_accumulation(renderer, total_sample_count=None):
    """Sums the accumulation targets of all ranks in place (every rank ends up with the full image)."""
    import torch
    import torch.distributed as dist
    renderer.synchronize()
    tensor = accumulation_tensor(renderer)
    if dist.is_initialized():
        # The alpha channel holds 1 on every rank; keep it 1 after the sum
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
        tensor[..., 3] = 1.0
    torch.cuda.synchronize()
    if total_sample_count is not None:
        from . import hip_api
        hip_api.lib().pt_set_sample_count(renderer.device_scene, total_sample_count)
    return tensor


def all_reduce_host(image, group=None):
    """CPU/gloo counterpart used by the tests: sums numpy images across ranks."""
    import torch
    import torch.distributed as dist
    tensor = torch.from_numpy(np.ascontiguousarray(image))
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    result = tensor.numpy()
    result[..., 3] = 1.0
    return result
