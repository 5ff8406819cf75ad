"""Renders a benchmark workload with the library as it is in the tree and writes the mean radiance as a .npy file
(for comparisons between builds of the library). usage: python scripts/gpu_render_to_file.py OUT.npy [workload] [spp] [width] [height]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from path_tracer_amd import host_api  # noqa: E402

out = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "city"
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
width = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
height = int(sys.argv[5]) if len(sys.argv) > 5 else 1080
frame_offset = int(sys.argv[6]) if len(sys.argv) > 6 else 0
args = bench.parse_args(["--workload", workload])
root = bench.workload_root(args, 0)
spec = host_api.quickload(os.path.join(root, "data", "saves", bench.WORKLOADS[workload][2]))
spec.frame_index += frame_offset
renderer = host_api.Renderer(spec, root, device=0, width=width, height=height)
done = 0
while done < spp:
    frames = min(256, spp - done)
    renderer.render_frames(frames)
    renderer.synchronize()
    done += frames
    print("%d of %d frames" % (done, spp), flush=True)
image, count = renderer.read_accumulation()
renderer.close()
np.save(out, (image[..., :3].astype(np.float64) / count).astype(np.float32))
