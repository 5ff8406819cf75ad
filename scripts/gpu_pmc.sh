#!/bin/bash
# Collects hardware counters for the benchmark kernels, one rocprofv3 --pmc pass per counter group
# (counters are not split over passes by the tool). usage: scripts/gpu_pmc.sh <tag> [bench args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
pass() {
	name=$1; shift
	counters=$1; shift
	timeout -s KILL 150 rocprofv3 --pmc $counters --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
	code=$?
	find $out/$name -name "*counter_collection.csv" -exec mv {} $out/$name.csv \;
	rm -rf $out/$name
	echo "pass $name exit $code"
	return $code
}
pass fetch "FETCH_SIZE" "$@" || exit 1
pass write "WRITE_SIZE" "$@" || exit 1
pass sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "$@" || exit 1
pass tcc "TCC_HIT_sum TCC_MISS_sum" "$@" || exit 1
# Where the waves spend their time: issuing, waiting for an issue slot, parked on a wait instruction (quad-cycles summed over waves)
pass sqwait "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "$@" || exit 1
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $out > $out/summary.txt
cat $out/summary.txt
