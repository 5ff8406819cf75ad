#!/bin/bash
# Benchmarks alternative builds of libpt_hip.so found as variants/libpt_hip_<name>.so (A/B of kernel variants).
# usage: scripts/gpu_variants.sh [bench args...]; build variants with scripts/build_variant.sh <name> <-D flags...>
cd $GRAFT_REPO_ROOT
cp path_tracer_amd/csrc/libpt_hip.so /tmp/libpt_hip_orig.so
for lib in variants/libpt_hip_*.so; do
	name=$(basename $lib .so)
	cp $lib path_tracer_amd/csrc/libpt_hip.so
	timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 8 --warmup 1 --strong-frames 0 "$@" > /tmp/bench_$name.json 2>/tmp/bench_$name.err || { echo "$name failed"; tail -3 /tmp/bench_$name.err; break; }
	python3 -c "
import json,sys; d=json.load(open('/tmp/bench_$name.json')); k=d['kernel_ms']; print('$name', round(d['value']), 'Mrays/s', round(d['ms_per_step'],1), 'ms/step', {n: round(v/8,2) for n,v in k.items()}, {n: (round(t['nodes_per_ray'],1), round(t['triangles_per_ray'],1)) for n,t in d.get('traversal',{}).items()})"
done
cp /tmp/libpt_hip_orig.so path_tracer_amd/csrc/libpt_hip.so
