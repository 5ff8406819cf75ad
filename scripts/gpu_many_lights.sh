#!/bin/bash
# Benchmarks the loop over all lights against the light tree for growing light counts on the generated city.
# usage: scripts/gpu_many_lights.sh [bench args...]
cd $GRAFT_REPO_ROOT
for lights in 28 1000 10000; do
	for mode in loop tree; do
		timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload city --steps 2 --warmup 1 --spp-per-step 8 --lights $lights --light-sampling $mode "$@" > /tmp/ml.json 2>/tmp/ml.err || { echo "$lights $mode failed"; tail -3 /tmp/ml.err; exit 1; }
		python3 -c "
import json; d=json.load(open('/tmp/ml.json')); k=d['kernel_ms']; print('lights $lights $mode:', round(d['value']), 'Mrays/s', round(d['ms_per_step'],1), 'ms/step', {n: round(v/d['steps'],2) for n,v in k.items()})"
	done
done
