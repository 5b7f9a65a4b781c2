#!/bin/bash
# sweep the SAH cost model on the GPU box: tools/sah_sweep.sh <workload>
for ct in 5; do for eb in 0.6 0.8 1.0; do for ml in 1 2 6; do
  echo "== traversal_cost $ct empty_bonus $eb min_leaf $ml"
  HXR_SAH_TRAVERSAL_COST=$ct HXR_SAH_EMPTY_BONUS=$eb HXR_SAH_MIN_LEAF=$ml timeout -k 10 200 python3 tools/tree_ab.py ${1:-meshes16} 2>&1 | grep " sah "
done; done; done
