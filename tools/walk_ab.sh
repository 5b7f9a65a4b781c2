#!/bin/bash
for accel in ref sah; do
  echo "== HXR_ACCEL=$accel"
  HXR_ACCEL=$accel timeout -k 10 200 python3 tools/run_frame.py meshes16 6
  HXR_ACCEL=$accel timeout -k 10 200 python3 tools/shard_time.py meshes16 | grep "world"
done
for w in simple1 heightfield16 smallpt16 cornell16 kdtree16; do python3 tools/run_frame.py $w 5; HXR_NO_ADAPTIVE_SCHEDULE=1 python3 tools/run_frame.py $w 5; done
