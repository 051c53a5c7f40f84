#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root:   bash profiles/collect_round.sh <tag>
# Everything a round's DESIGN numbers are taken from, under gpurun_out/<tag>/ :
#   bench.json                    python bench.py (default arguments: the driver's command)
#   collect/...                   profiles/collect.sh: rocprofv3 kernel stats of the bench step + PMC passes + pmc_traffic.json
#   c3_kernel_stats.csv           rocprofv3 --kernel-trace --stats of three fresh config-3 constructions (profiles/c3_pmc_run.py)
#   timeline/timeline_summary.json  rocprofv3 --hip-trace --kernel-trace of ten steps (per-kernel Gantt, idle gaps)
#   host_stamps.txt               XR_HOST_STAMPS=1: where the host's time goes along a weight build
set -u
TAG=${1:-round}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
bash profiles/collect.sh "$TAG/collect" > "$OUT/collect.log" 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3" -o c3 -- python "$ROOT/profiles/c3_pmc_run.py" > "$OUT/c3.log" 2>&1 )
cp "$(find "$OUT/c3" -name '*kernel_stats.csv' | head -1)" "$OUT/c3_kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/c3"
bash profiles/timeline.sh "$TAG/timeline" 10 > /dev/null 2>&1
XR_HOST_STAMPS=1 python profiles/timeline_run.py 200 2> "$OUT/host_stamps.txt" > /dev/null
find "$OUT" -name "*.csv" -size +8M -delete
ls -la "$OUT" "$OUT/collect" | head -40
