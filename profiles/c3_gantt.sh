#!/bin/bash
# ON THE GPU BOX: bash profiles/c3_gantt.sh <tag>  -> gpurun_out/<tag>/c3_gantt.json (last fresh and last cached construction)
set -u
TAG=${1:-c3g}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o c3 -- python "$ROOT/profiles/c3_gantt_run.py" > "$OUT/run.log" 2>&1
cd "$ROOT"
python profiles/c3_gantt.py "$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)" > "$OUT/c3_gantt.json"
rm -rf "$OUT/trace"; tail -8 "$OUT/run.log"
