#!/bin/bash
# Everything profiles/ holds for a round, in one gpurun call: bench lines of every workload, the rocprof / PMC profile of the headline and of
# the other config kernels, the 8-way shard bounds, phase clocks, the interactive loop, scene preparation.  usage: tools/final_round.sh <tag>
tag=${1:-r03}
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python3 bench.py > $out/bench.json 2> $out/bench.err
: > $out/bench_all_workloads.jsonl
for w in meshes16 meshes_aa simple1 heightfield16 smallpt16 smallpt256 cornell16 kdtree16; do
	timeout -k 10 300 python3 bench.py --workload $w --steps 10 --warmup 2 --no-other-timings >> $out/bench_all_workloads.jsonl 2>> $out/bench.err
done
timeout -k 10 300 python3 bench.py --accel sah --no-other-timings > $out/bench_sah.json 2>> $out/bench.err
tools/profile_round.sh $tag > $out/profile.log 2>&1
for w in heightfield16 cornell16 smallpt16; do tools/pmc.sh $w gpurun_out/$tag/pmc_$w > $out/pmc_$w.txt 2>&1; rm -rf $out/pmc_$w/pass*/; done
HXR_TILE=8 HXR_ALL_RANKS=1 timeout -k 10 300 python3 tools/shard_time.py meshes16 > $out/shard_times_reference_tree.txt 2>&1
HXR_ACCEL=sah HXR_TILE=8 HXR_ALL_RANKS=1 timeout -k 10 300 python3 tools/shard_time.py meshes16 > $out/shard_times_sah_tree.txt 2>&1
HXR_TILE=8 HXR_ALL_RANKS=1 timeout -k 10 300 python3 tools/shard_time.py smallpt256 > $out/shard_times_smallpt256.txt 2>&1
timeout -k 10 300 python3 tools/sah_build_times.py > $out/sah_build_times.txt 2>&1
timeout -k 10 120 hexray_amd/bin/hexray data/simple.hexray --interactive 300 > $out/interactive_simple.txt 2>&1
timeout -k 10 120 hexray_amd/bin/hexray data/simple.hexray --interactive 300 --fovea 75 >> $out/interactive_simple.txt 2>&1
timeout -k 10 120 hexray_amd/bin/hexray data/meshes.hexray --width 1920 --height 1080 --dof on --spp 16 --frames 5 --devices 0,0 >> $out/interactive_simple.txt 2>&1
# round 4: the lone chain's split (needs tools/build_alt.sh chain4 -DHXR_PHASE_TIMERS -DHXR_PHASE_SET=4 beforehand), the adaptive-AA frame's per-kernel timeline,
# and the A/B lines of the latency measures (priorities, replicating blocks) on the headline's three figures
[ -f hexray_amd/lib/libhexray_hip_chain4.so ] && bash tools/r4_chain.sh $tag > $out/chain.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$out/aa_trace -- python3 $OLDPWD/tools/run_frame.py meshes_aa 6 > $OLDPWD/$out/aa_trace.log 2>&1)
find $out/aa_trace -name "*kernel_trace.csv" -exec cp {} $out/aa_kernel_trace.csv \; ; rm -rf $out/aa_trace
python3 tools/aa_timeline.py $out/aa_kernel_trace.csv > $out/aa_timeline.txt 2>&1
bash tools/r4_ab2.sh $tag "-|A=1" "-|HXR_NO_PARTS_KERNEL=1" "-|HXR_NO_PARTS_KERNEL=1 HXR_NO_FRONT_PRIORITY=1 HXR_NO_FIRST_PRIORITY=1" > /dev/null 2>&1
# when every wave of the headline frame starts and leaves, steady state against the first frame of a fresh handle, and for one rank of the 8-way frame
# (needs tools/build_alt.sh wt -DHXR_WAVE_TIMELINE beforehand)
if [ -f hexray_amd/lib/libhexray_hip_wt.so ]; then
	HXR_HIP_LIB=$PWD/hexray_amd/lib/libhexray_hip_wt.so timeout -k 10 200 python3 tools/wave_timeline.py 2>&1 | grep -v "Warn\|setattr\|return self\|amdgpu.ids" > $out/wave_timeline.txt
	HXR_SHARD=0,8 HXR_HIP_LIB=$PWD/hexray_amd/lib/libhexray_hip_wt.so timeout -k 10 200 python3 tools/wave_timeline.py 2>&1 | grep -v "Warn\|setattr\|return self\|amdgpu.ids" | tail -6 >> $out/wave_timeline.txt
fi
timeout -k 10 100 python3 tools/exp_shard_period.py 8 2>&1 | grep period > $out/shard_period.txt
# the sharded pipeline end to end with four ranks on the one GPU (gloo gather; NOT a measurement): the assembled frame must equal the single-GPU frame
HXR_BENCH_REHEARSAL=1 HXR_BENCH_VERIFY=1 timeout -k 10 400 python3 bench.py --gpus 4 --steps 3 --warmup 1 2> $out/bench_rehearsal.err | grep "^{" > $out/bench_rehearsal_4_ranks_one_gpu.json
tail -2 $out/bench.json | cut -c1-600; tail -8 $out/interactive_simple.txt; cat $out/ab.txt; tail -c 400 $out/bench_rehearsal_4_ranks_one_gpu.json; grep -i identical $out/bench_rehearsal.err
