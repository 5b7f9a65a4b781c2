#!/bin/bash
# Copy what tools/final_round.sh <tag> left under gpurun_out/<tag>/ into profiles/ (the judged summaries) and refresh profiles/hbm_traffic.json /
# pmc_compute.json for the four profiled workloads.  usage: tools/collect_round.sh r03
tag=${1:-r03}; src=gpurun_out/$tag
python3 tools/profile_collect.py $tag meshes16 > /dev/null
for f in bench.json bench_all_workloads.jsonl bench_sah.json shard_times_reference_tree.txt shard_times_sah_tree.txt shard_times_smallpt256.txt sah_build_times.txt; do cp $src/$f profiles/${tag}_$f; done
cp $src/interactive_simple.txt profiles/${tag}_interactive_and_group.txt
for w in heightfield16 cornell16 smallpt16; do
	mkdir -p gpurun_out/${tag}_$w && rm -rf gpurun_out/${tag}_$w/pmc && cp -r $src/pmc_$w gpurun_out/${tag}_$w/pmc && python3 tools/profile_collect.py ${tag}_$w $w > /dev/null
done
# round 4 additions (when present): the lone chain's split, the adaptive-AA timeline, the A/B lines of the latency measures
for f in chain_reference.txt chain_sah.txt aa_timeline.txt ab.txt bench_rehearsal_4_ranks_one_gpu.json wave_timeline.txt shard_period.txt; do [ -f $src/$f ] && cp $src/$f profiles/${tag}_$f; done
[ -f $src/ab.txt ] && mv profiles/${tag}_ab.txt profiles/${tag}_latency_measures_ab.txt
true
