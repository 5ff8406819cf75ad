#!/bin/bash
# What the one-tap texture filter leaves out against an anisotropic filter of up to 16 taps (the reference's sampler:
# maxAnisotropy = 16, src/main.c:864-873): the generated city at 1920x1080 and 2048 samples per pixel with the library
# as it is and with variants/libpt_hip_taps16.so (scripts/build_variant.sh taps16 -DPT_TEXTURE_TAPS=16), equal seeds,
# and a second run of the one-tap library with other seeds for the noise floor. Writes gpurun_out/taps/result.txt.
cd $GRAFT_REPO_ROOT
spp=${1:-2048}
mkdir -p gpurun_out/taps
cp path_tracer_amd/csrc/libpt_hip.so /tmp/libpt_hip_orig.so
timeout -k 10 500 python3 -u scripts/gpu_render_to_file.py /tmp/one_tap.npy city $spp 1920 1080 0 > gpurun_out/taps/one_tap.log 2>&1 || { tail -3 gpurun_out/taps/one_tap.log; exit 1; }
timeout -k 10 500 python3 -u scripts/gpu_render_to_file.py /tmp/one_tap_other_seeds.npy city $spp 1920 1080 100000 > gpurun_out/taps/one_tap_other.log 2>&1 || { tail -3 gpurun_out/taps/one_tap_other.log; exit 1; }
cp variants/libpt_hip_taps16.so path_tracer_amd/csrc/libpt_hip.so
timeout -k 10 900 python3 -u scripts/gpu_render_to_file.py /tmp/taps16.npy city $spp 1920 1080 0 > gpurun_out/taps/taps16.log 2>&1
code=$?
cp /tmp/libpt_hip_orig.so path_tracer_amd/csrc/libpt_hip.so
[ $code -eq 0 ] || { tail -3 gpurun_out/taps/taps16.log; exit 1; }
python3 - $spp <<'PY' | tee gpurun_out/taps/result.txt
import sys
import numpy as np
spp = int(sys.argv[1])
one, other, taps = (np.load(p).astype(np.float64) for p in ("/tmp/one_tap.npy", "/tmp/one_tap_other_seeds.npy", "/tmp/taps16.npy"))
def relative_rmse(a, b):
    return float(np.sqrt(((a - b) ** 2).mean()) / b.mean())
changed = (one != taps).any(axis=2).mean()
print("generated city 1920x1080, %d samples per pixel, NEE, path length 4" % spp)
print("one trilinear tap against up to 16 taps along the major axis, equal seeds: relative RMSE %.5f, mean radiance %.5f vs %.5f (%.3f %% apart), %.1f %% of the pixels differ"
      % (relative_rmse(one, taps), one.mean(), taps.mean(), 100.0 * abs(one.mean() / taps.mean() - 1.0), 100.0 * changed))
print("one tap against one tap with other seeds (the noise at this sample count): relative RMSE %.5f" % relative_rmse(one, other))
# per-channel and on a blurred image, where the noise of differing paths averages out
def box(a, k=8):
    h, w, c = a.shape
    return a[:h - h % k, :w - w % k].reshape(h // k, k, w // k, k, c).mean(axis=(1, 3))
print("the same on 8x8 pixel averages: one tap vs 16 taps %.5f, one tap vs other seeds %.5f" % (relative_rmse(box(one), box(taps)), relative_rmse(box(one), box(other))))
PY
