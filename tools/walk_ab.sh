#!/bin/bash
for div in 2 1 4; do
  export HXR_SCHED_FRAME_DIV=$div
  echo "== frame_div $div"
  python3 tools/run_frame.py meshes16 6
  HXR_ACCEL=sah python3 tools/run_frame.py meshes16 6
  python3 tools/run_frame.py heightfield16 5
  HXR_ALL_RANKS=1 HXR_TILE=8 timeout -k 10 300 python3 tools/shard_time.py meshes16 | grep "world 8\|world 4" | awk '{print $1,$2,$3,$4, $NF, $(NF-4)}' | tr '\n' ';'; echo
done
