#!/bin/bash
# Per-launch durations of the traversal kernels of one bench run (kernel trace), to compare bounces.
# usage: scripts/gpu_per_bounce.sh [bench args...]; writes gpurun_out/per_bounce/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/per_bounce && mkdir -p gpurun_out/per_bounce
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/per_bounce -o trace -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > gpurun_out/per_bounce/bench.json 2> gpurun_out/per_bounce/bench.err || { tail -5 gpurun_out/per_bounce/bench.err; exit 1; }
python3 - <<'PY'
import csv, glob, json
path = glob.glob('gpurun_out/per_bounce/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(path)) if r['Kernel_Name'].startswith(('void k_trace', 'k_trace', 'void k_shade', 'k_shade'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# The last step: the kernels after the last raygen
out = []
for r in rows[-40:]:
    out.append((r['Kernel_Name'][:40], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
for name, ms in out[-12:]:
    print('%-42s %8.3f ms' % (name, ms))
print(open('gpurun_out/per_bounce/bench.json').read()[:600])
PY
