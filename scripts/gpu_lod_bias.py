"""How far the level-of-detail rule (ray cones, trilinear) moves a converged image: the generated city rendered with
its mip chains against the same scene whose textures keep level 0 only (every lookup bilinear at full resolution,
which with the pixel jitter converges to the unfiltered ground truth), and both against a second run with other frame
indices (the noise floor at that sample count). Prints relative RMSE figures, from which the
tolerance against a driver's filter is stated. usage: python scripts/gpu_lod_bias.py [spp] [width] [height]"""
import glob
import os
import shutil
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from path_tracer_amd import host_api  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
width = int(sys.argv[2]) if len(sys.argv) > 2 else 480
height = int(sys.argv[3]) if len(sys.argv) > 3 else 270
sys.argv = [sys.argv[0], "--workload", "city"]
args = bench.parse_args()
root = bench.workload_root(args, 0)
flat_root = root + "_level0"
if not os.path.exists(flat_root):
    shutil.copytree(root, flat_root)
    for path in glob.glob(os.path.join(flat_root, "data", "*_textures", "*.vkt")):
        data = open(path, "rb").read()
        marker, version, mips, w, h, fmt, payload = struct.unpack_from("<IIIIIIQ", data, 0)
        mip_w, mip_h, size, offset = struct.unpack_from("<IIQQ", data, 32)
        begin = 32 + 24 * mips + offset
        with open(path, "wb") as out:
            out.write(struct.pack("<IIIIIIQ", marker, version, 1, w, h, fmt, size))
            out.write(struct.pack("<IIQQ", mip_w, mip_h, size, 0))
            out.write(data[begin:begin + size])
            out.write(struct.pack("<I", 0xe0fe0f))
save = bench.WORKLOADS["city"][2]


def render(scene_root, frame_offset):
    spec = host_api.quickload(os.path.join(scene_root, "data", "saves", save))
    spec.frame_index += frame_offset
    renderer = host_api.Renderer(spec, scene_root, device=0, width=width, height=height)
    renderer.render_frames(spp)
    image, count = renderer.read_accumulation()
    renderer.close()
    return image[..., :3].astype(np.float64) / count


with_mips = render(root, 0)
with_mips_again = render(root, 100000)
level_0 = render(flat_root, 0)
level_0_again = render(flat_root, 100000)


def relative_rmse(a, b):
    return float(np.sqrt(((a - b) ** 2).mean()) / b.mean())


noise = relative_rmse(with_mips, with_mips_again)
noise_0 = relative_rmse(level_0, level_0_again)
difference = relative_rmse(with_mips, level_0)
print("city %dx%d, %d spp: relative RMSE between two runs with mip chains %.4f, between two runs with level 0 only %.4f, "
      "between mip chains and level 0 at equal seeds %.4f (the noise of equal paths cancels: this bounds what the rule changes); "
      "mean radiance %.4f vs %.4f (%.2f %% apart)"
      % (width, height, spp, noise, noise_0, difference, with_mips.mean(), level_0.mean(), 100.0 * abs(with_mips.mean() / level_0.mean() - 1.0)), file=sys.stderr)
