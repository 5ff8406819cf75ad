#!/bin/bash
# The profile set of a round on one GPU box: kernel statistics (rocprofv3 --kernel-trace --stats) and the counter passes
# (separate rocprofv3 --pmc runs) of the default benchmark command for the city and the Cornell box, then plain bench
# lines of every configuration. usage: scripts/gpu_round_profiles.sh <tag>; results under gpurun_out/, to be copied
# into profiles/ with scripts/collect_profiles.sh <tag>.
tag=$1
cd $GRAFT_REPO_ROOT
common="--steps 8 --warmup 1 --strong-frames 0"
for w in city cornell; do
	scripts/gpu_profile.sh ${w}_$tag --workload $w $common > gpurun_out/profile_${w}_$tag.log 2>&1 || { echo "profile $w failed"; tail -5 gpurun_out/profile_${w}_$tag.log; exit 1; }
	echo "profile $w done"
	scripts/gpu_pmc.sh ${w}_$tag --workload $w $common > gpurun_out/pmc_${w}_$tag.log 2>&1 || { echo "pmc $w failed"; tail -5 gpurun_out/pmc_${w}_$tag.log; exit 1; }
	echo "pmc $w done"
done
timeout -k 10 300 python3 bench.py --workload city > gpurun_out/bench_city_$tag.json 2> gpurun_out/bench_city_$tag.err && echo "bench city done" || exit 1
timeout -k 10 300 python3 bench.py --workload cornell > gpurun_out/bench_cornell_$tag.json 2> gpurun_out/bench_cornell_$tag.err && echo "bench cornell done" || exit 1
timeout -k 10 300 python3 bench.py --workload city --shading-math fast --no-cpu-baseline > gpurun_out/bench_city_fast_$tag.json 2> gpurun_out/bench_city_fast_$tag.err && echo "bench city fast done" || exit 1
timeout -k 10 300 python3 bench.py --workload cornell --width 512 --height 512 --spp-per-step 64 --steps 16 --no-cpu-baseline > gpurun_out/bench_c1_$tag.json 2> gpurun_out/bench_c1_$tag.err && echo "bench c1 done" || exit 1
timeout -k 10 300 python3 bench.py --workload interior --width 3840 --height 2160 --spp-per-step 8 --steps 8 --strong-frames 0 --no-cpu-baseline > gpurun_out/bench_c4_$tag.json 2> gpurun_out/bench_c4_$tag.err && echo "bench c4 done" || exit 1
