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
# The vector memory path: how long the address units (one per CU) are busy and stalled by the vector L1, how long the L1
# stalls on outstanding misses, its requests to L2 and their latency (cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
pass ta "GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" "$@" || exit 1
pass ta2 "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" "$@" || exit 1
pass tcc2 "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "$@" || exit 1
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $out > $out/summary.txt
cat $out/summary.txt
