#!/bin/bash
# One SQ counter pass over a short benchmark run. usage: scripts/gpu_pmc_sq.sh <tag> [bench args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/sq -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/sq.json 2> $out/sq.err
find $out/sq -name "*counter_collection.csv" -exec mv {} $out/sq.csv \;
rm -rf $out/sq
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $out | grep -E "k_trace|kernel,"
