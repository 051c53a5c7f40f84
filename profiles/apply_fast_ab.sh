#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root:  bash profiles/apply_fast_ab.sh <tag>
# The price of bit-exactness in the many-variable apply (BASELINE config 5), as profiles/fma_ab.sh did for the clip:
# K = 256 on the qhull-numbered benchmark matrix and on the lattice-numbered pair, exact (default) against contracted
# (XR_APPLY_CONTRACT=1), each also with gathers and stores switched off (XR_PLAN_DBG=3: the reduction's floor); and the accuracy
# of both against the CPU oracle at full size.  -> gpurun_out/<tag>/apply_fast_ab.txt
set -u
TAG=${1:-fastab}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
{
for KIND in delaunay lattice; do
    echo "== $KIND: K = 256 apply, 10 back-to-back"
    python profiles/apply_sweep.py $KIND XR_APPLY_CONTRACT=1 XR_APPLY_CONTRACT=0 XR_APPLY_CONTRACT=1 XR_PLAN_DBG=3 XR_PLAN_DBG=3,XR_APPLY_CONTRACT=1
    echo "== $KIND: accuracy at full size (K = 16)"
    python profiles/apply_fast_accuracy.py $KIND 16
done
} > "$OUT/apply_fast_ab.txt" 2>&1
cat "$OUT/apply_fast_ab.txt"
