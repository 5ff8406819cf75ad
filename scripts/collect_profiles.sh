#!/bin/bash
# Copies the profiling results of scripts/gpu_profile.sh and scripts/gpu_pmc.sh runs tagged cornell_<tag> and city_<tag>
# from gpurun_out/ into profiles/<round>/<tag>/ and rebuilds profiles/pmc_current.json. usage: scripts/collect_profiles.sh <tag> [round, default r3]
set -e
tag=$1
round=${2:-r3}
dst=profiles/$round/$tag
mkdir -p $dst
for w in cornell city; do
	cp gpurun_out/pmc_${w}_$tag/summary.txt $dst/pmc_${w}_summary.txt
	cp gpurun_out/prof_${w}_$tag/kernel_stats.csv $dst/kernel_stats_$w.csv
	cp gpurun_out/prof_${w}_$tag/bench.json $dst/bench_${w}_under_rocprof.json
	[ -f gpurun_out/bench_${w}_$tag.json ] && cp gpurun_out/bench_${w}_$tag.json $dst/
done
rm -f profiles/pmc_current.json
for w in cornell city; do
	python3 scripts/pmc_to_json.py $dst/pmc_${w}_summary.txt $w profiles/pmc_current.json $dst/kernel_stats_$w.csv > /dev/null
done
