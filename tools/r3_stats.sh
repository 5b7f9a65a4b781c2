#!/bin/bash
# round 3, first look: baseline kernel times + the developer counters of the stats builds (one gpurun call)
out=gpurun_out/r3a; mkdir -p $out
for w in meshes16 heightfield16 cornell16 smallpt16; do timeout -k 10 200 python3 tools/run_frame.py $w 5 >> $out/base.txt 2>&1 || echo "failed $w" >> $out/base.txt; done
for tag in leafstats leafhist walkstats; do
  HXR_HIP_LIB=hexray_amd/lib/libhexray_hip_$tag.so timeout -k 10 300 python3 tools/walk_stats.py meshes16 cornell16 >> $out/stats.txt 2>&1 || echo "failed $tag" >> $out/stats.txt
done
cat $out/base.txt $out/stats.txt
