#!/bin/bash
# ON THE GPU BOX: bash profiles/apply_k256_gantt.sh <tag> -> gpurun_out/<tag>/apply_gantt.txt: kernels of the last K = 256 applies with start / end
set -u
TAG=${1:-ag}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o ap -- python "$ROOT/profiles/apply_k256_run.py" delaunay 256 4 > "$OUT/run.log" 2>&1
cd "$ROOT"
python - "$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)" > "$OUT/apply_gantt.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ker = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xr::", "")[:40], r.get("Queue_Id", "0")) for r in rows), key=lambda x: x[0])
ker = ker[-24:]
t0 = ker[0][0]
for s, e, n, q in ker:
    print(f"{n:40s} q{q} {(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:.1f})")
PY
rm -rf "$OUT/trace"; tail -2 "$OUT/run.log"; cat "$OUT/apply_gantt.txt"
