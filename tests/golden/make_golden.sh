#!/bin/bash
# Regenerates tests/golden/reference_*_vectors.json from the reference's own code (needs /root/reference).
set -e
cd "$(dirname "$0")/../.."
make -s -C oracle ref
oracle/_ref/ref_harness > tests/golden/reference_vectors.json
oracle/_ref/ref_harness_shading > tests/golden/reference_shading_vectors.json
oracle/_ref/ref_harness_shading diffuse > tests/golden/reference_diffuse_vectors.json   # Lambert / Phong over the reference's own lights (a process of its own: fresh generator)
for f in reference_vectors reference_shading_vectors reference_diffuse_vectors; do
python3 -c "import json; d=json.load(open('tests/golden/$f.json')); print({k:(len(v) if isinstance(v,list) else v) for k,v in d.items()})"
done
