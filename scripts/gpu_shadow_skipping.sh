#!/bin/bash
# Shadow rays with and without skipping of what cannot change the answer (PT_SHADOW_SKIPPING=0, 1, unset = decided
# by the rays traced so far), on several workloads. usage: scripts/gpu_shadow_skipping.sh
cd $GRAFT_REPO_ROOT
for workload in "--workload city" "--workload cornell" "--workload interior --spp-per-step 8" "--workload cornell --width 512 --height 512 --spp-per-step 64"; do
	for mode in 0 1 auto; do
		if [ $mode = auto ]; then unset PT_SHADOW_SKIPPING; else export PT_SHADOW_SKIPPING=$mode; fi
		PT_TRACE_DIAGNOSTICS=1 timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 $workload > /tmp/skip.json 2>/tmp/skip.err || { echo "$workload $mode failed"; tail -3 /tmp/skip.err; exit 1; }
		python3 -c "
import json
line=[l for l in open('/tmp/skip.json') if l.startswith('{')][-1]
d=json.loads(line); k=d['kernel_ms']; t=d.get('traversal',{}).get('shadow',{})
print('$workload', 'skipping=$mode', round(d['value']), 'Mrays/s, shadow', round(k['trace_shadow']/d['steps'],2), 'ms/step, nodes per ray', round(t.get('nodes_per_ray',0),1))"
		grep -h "ended at a hit" /tmp/skip.json /tmp/skip.err | tail -1
	done
done
