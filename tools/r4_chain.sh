#!/bin/bash
# Round 4, VERDICT r3 #4: what does ONE wave's dependent chain wait for?  The slowest 8x8 tile of the headline frame alone on the GPU (one packet's chain IS the
# kernel time), through the chain-profile build (tools/build_alt.sh chain4 -DHXR_PHASE_TIMERS -DHXR_PHASE_SET=4): reference tree and SAH tree.
out=gpurun_out/${1:-r4b}; mkdir -p $out
names="outside Mesh::intersect,kd compute (box tests + ballots + pushes),kd wait (children / popped record),leaf wait (hot record),leaf compute,pop"
for accel in reference sah; do
	HXR_ACCEL=$accel timeout -k 10 300 python3 tools/exp_chain.py > $out/chain_$accel.txt 2>&1 || exit 1
	t8=$(grep -o 'HXR_TILE8=[0-9]*' $out/chain_$accel.txt | tail -1 | cut -d= -f2)
	HXR_ACCEL=$accel HXR_TILE8=$t8 HXR_PHASE_NAMES="$names" HXR_HIP_LIB=$(pwd)/hexray_amd/lib/libhexray_hip_chain4.so timeout -k 10 120 python3 tools/exp_chain.py >> $out/chain_$accel.txt 2>&1 || exit 1
	tail -4 $out/chain_$accel.txt
done
