#!/bin/bash
# Throughput against the number of frames per pt_render_samples() call. usage: scripts/gpu_spp_sweep.sh [bench args...]
for spp in 1 2 4 8 32; do
	timeout -k 10 200 python bench.py --no-cpu-baseline --steps 32 --spp-per-step $spp "$@" 2>/dev/null | SPP=$spp python3 -c "
import json, os, sys
d = json.loads(sys.stdin.read())
print('frames per call', os.environ['SPP'], round(d['value']), 'Mrays/s', round(d['ms_per_step'], 3), 'ms per call', {k: round(v / 32, 3) for k, v in d['kernel_ms'].items()})" || exit 1
done
