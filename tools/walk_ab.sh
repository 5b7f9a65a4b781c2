#!/bin/bash
for tile in 32 16 8; do
  echo "== tile $tile"
  HXR_ALL_RANKS=1 HXR_TILE=$tile timeout -k 10 300 python3 tools/shard_time.py meshes16 | grep "world 8\|world 1"
done
