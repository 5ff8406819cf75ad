#!/bin/bash
# Profiles the benchmark with rocprofv3 on the GPU box; summaries land in gpurun_out/prof_<tag>/.
# usage: scripts/gpu_profile.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err || exit 1
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out/trace -name "*kernel_trace.csv" -exec sh -c 'head -1 "$1" > '$out'/kernel_trace_head.csv; wc -l "$1" >> '$out'/kernel_trace_head.csv' _ {} \;
rm -rf $out/trace
cat $out/kernel_stats.csv | cut -c1-200
