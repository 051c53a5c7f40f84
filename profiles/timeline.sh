#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root:   bash profiles/timeline.sh <tag> [steps]
# rocprofv3 --hip-trace --kernel-trace (no counters: gpurun refuses the combination) of profiles/timeline_run.py, then the
# summary: gpurun_out/<tag>/timeline_summary.json
set -u
TAG=${1:-timeline}; STEPS=${2:-10}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --output-format csv -d "$OUT/trace" -o tl -- python "$ROOT/profiles/timeline_run.py" "$STEPS" > "$OUT/run.log" 2>&1
cd "$ROOT"
K=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
H=$(find "$OUT/trace" -name '*hip_api_trace.csv' | head -1)
python profiles/timeline_summary.py "$K" "$H" "$STEPS" > "$OUT/timeline_summary.json"
# the raw traces are large: keep only the summary and a head of each
head -3 "$K" > "$OUT/kernel_trace_head.csv"; head -3 "$H" > "$OUT/hip_api_trace_head.csv"
rm -rf "$OUT/trace"
