"""Renders the benchmark's scenes to a thousand frames and counts the pixels whose sum is not finite or negative: the
reference guards neither (a density of zero makes a throughput infinite, pathtrace.frag.glsl:304,355), so such pixels
are its behaviour where they occur, and this run says how often that is.
usage (on a GPU box): python scripts/gpu_image_sanity.py > gpurun_out/image_sanity.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from path_tracer_amd import host_api  # noqa: E402

for workload, width, height, spp in (("cornell", 1920, 1080, 1024), ("city", 1920, 1080, 1024), ("interior", 3840, 2160, 128)):
    args = bench.parse_args(["--workload", workload])
    root = bench.workload_root(args, 0)
    spec = host_api.quickload(os.path.join(root, "data", "saves", bench.WORKLOADS[workload][2]))
    renderer = host_api.Renderer(spec, root, device=0, width=width, height=height)
    done = 0
    while done < spp:
        frames = min(128, spp - done)
        renderer.render_frames(frames)
        renderer.synchronize()
        done += frames
    image, count = renderer.read_accumulation()
    renderer.close()
    rgb = image[..., :3]
    not_finite = ~np.isfinite(rgb).all(axis=2)
    negative = (rgb < 0.0).any(axis=2)
    finite = rgb[~not_finite]
    print("%s %dx%d, %d frames: %d of %d pixels not finite, %d negative, alpha %s, mean of the others %.5f, largest %.1f"
          % (workload, width, height, count, int(not_finite.sum()), width * height, int(negative.sum()),
             "all one" if (image[..., 3] == 1.0).all() else "NOT all one", float(finite.mean()) / count, float(finite.max()) / count), flush=True)
    for y, x in np.argwhere(not_finite | negative)[:5]:
        print("   pixel (%d, %d): %s" % (x, y, rgb[y, x]), flush=True)
