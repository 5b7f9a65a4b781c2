#!/bin/bash
# sweep the SAH cost model on the GPU box: tools/sah_sweep.sh <workload>
for ct in 3 4 5 6 8; do
  echo "== traversal_cost $ct"
  HXR_NO_ADAPTIVE_SCHEDULE=1 HXR_SAH_TRAVERSAL_COST=$ct timeout -k 10 200 python3 tools/tree_ab.py ${1:-meshes16} 2>&1 | grep " sah "
done
