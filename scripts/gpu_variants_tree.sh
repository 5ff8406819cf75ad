#!/bin/bash
# A/B of library variants (variants/libpt_hip_<name>.so) on the many-lights configuration of bench.py.
cd $GRAFT_REPO_ROOT
cp path_tracer_amd/csrc/libpt_hip.so /tmp/libpt_hip_orig.so
for lib in variants/libpt_hip_*.so; do
	name=$(basename $lib .so)
	cp $lib path_tracer_amd/csrc/libpt_hip.so
	timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload city --steps 2 --warmup 1 --spp-per-step 8 --lights 10000 --light-sampling tree "$@" > /tmp/bench_$name.json 2>/tmp/bench_$name.err || { echo "$name failed"; tail -3 /tmp/bench_$name.err; continue; }
	python3 -c "
import json; d=json.load(open('/tmp/bench_$name.json')); k=d['kernel_ms']; print('$name', round(d['value']), 'Mrays/s', round(d['ms_per_step'],1), 'ms/step', {n: round(v/d['steps'],2) for n,v in k.items()})"
done
cp /tmp/libpt_hip_orig.so path_tracer_amd/csrc/libpt_hip.so
