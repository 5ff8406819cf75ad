#!/bin/bash
# One counter pass with a caller-chosen counter list. usage: scripts/gpu_pmc_custom.sh <tag> "<counters>" [bench args...]
tag=$1; shift
counters=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d $out/c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/c.json 2> $out/c.err
find $out/c -name "*counter_collection.csv" -exec mv {} $out/c.csv \;
rm -rf $out/c
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $out | grep -E "k_trace_[a-z]*<false>|k_shade|kernel,"
tail -2 $out/c.err | cut -c1-200
