#!/bin/bash
# BVH build time of library variants (variants/libpt_hip_<name>.so) on a generated workload. usage: scripts/gpu_variants_build.sh [bench args...]
cd $GRAFT_REPO_ROOT
cp path_tracer_amd/csrc/libpt_hip.so /tmp/libpt_hip_orig.so
for lib in variants/libpt_hip_*.so; do
	name=$(basename $lib .so)
	cp $lib path_tracer_amd/csrc/libpt_hip.so
	timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp-per-step 4 "$@" > /tmp/bench_$name.json 2>/tmp/bench_$name.err || { echo "$name failed"; tail -3 /tmp/bench_$name.err; continue; }
	python3 -c "
import json; d=json.load(open('/tmp/bench_$name.json')); print('$name', d['bvh'], round(d['value']), 'Mrays/s')"
done
cp /tmp/libpt_hip_orig.so path_tracer_amd/csrc/libpt_hip.so
