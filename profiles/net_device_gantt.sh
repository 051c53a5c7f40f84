#!/bin/bash
# ON THE GPU BOX: bash profiles/net_device_gantt.sh <tag> -> gpurun_out/<tag>/net_device_gantt.txt (kernels of the last device-part call)
set -u
TAG=${1:-ngd}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o ng -- python "$ROOT/profiles/net_device_run.py" > "$OUT/run.log" 2>&1
cd "$ROOT"
python - "$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)" > "$OUT/net_device_gantt.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ker = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xr::", "")[:40]) for r in rows)
# the last call: from the last k_edge_tile_count on
last = max(i for i, k in enumerate(ker) if k[2].startswith("k_edge_tile_count"))
seg = ker[last - 1:] if last > 0 and "fillBuffer" in ker[last - 1][2] else ker[last:]
t0 = seg[0][0]
prev = None
for s, e, n in seg:
    gap = "" if prev is None else f"   gap {(s - prev) / 1e3:6.1f}"
    print(f"{n:42s} {(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f}){gap}")
    prev = e if prev is None else max(prev, e)
PY
rm -rf "$OUT/trace"; cat "$OUT/net_device_gantt.txt"
