#!/bin/bash
# Counters of the vector memory path (address units, vector L1, L2 latency) for the benchmark kernels,
# one rocprofv3 --pmc pass per group. usage: scripts/gpu_pmc_mem.sh <tag> [bench args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_mem_$tag
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
pass ta "GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" "$@" || exit 1
pass ta2 "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" "$@" || exit 1
pass tcc2 "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "$@" || exit 1
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $out > $out/summary.txt
grep -E "k_trace|k_shade" $out/summary.txt
