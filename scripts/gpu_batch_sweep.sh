#!/bin/bash
# Sweeps the number of paths per wavefront batch. usage: scripts/gpu_batch_sweep.sh [bench args...]
for paths in 4194304 8388608 16777216 33554432 67108864; do
	echo "PT_BATCH_PATHS=$paths"
	PT_BATCH_PATHS=$paths timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(' Mrays/s %.0f  ms/step %.2f ' % (d['value'], d['ms_per_step']), {k: round(v,1) for k,v in d['kernel_ms'].items()})" || exit 1
done
