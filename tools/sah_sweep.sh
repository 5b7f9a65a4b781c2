#!/bin/bash
# sweep the SAH cost model on the GPU box: tools/sah_sweep.sh <workload>
for ct in 1 2 3 4 6 8 12; do for eb in 0.8; do
  echo "== traversal_cost $ct empty_bonus $eb"
  HXR_SAH_TRAVERSAL_COST=$ct HXR_SAH_EMPTY_BONUS=$eb timeout -k 10 200 python3 tools/tree_ab.py ${1:-meshes16} 2>&1 | grep " sah "
done; done
