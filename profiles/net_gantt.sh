#!/bin/bash
set -u
TAG=${1:-ng}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -o ng -- python "$ROOT/profiles/net_gantt_run.py" > "$OUT/run.log" 2>&1
cd "$ROOT"
python - "$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)" "$(find "$OUT/trace" -name '*memory_copy_trace.csv' | head -1)" > "$OUT/net_gantt.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ker = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xr::", "")[:44], "q" + r.get("Queue_Id", "0")) for r in rows]
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:30], "dma"))
except Exception as e:
    print("no copy trace", e)
ker.sort()
segs, cur = [], []
for k in ker:
    if cur and k[0] - max(c[1] for c in cur) > 5_000_000: segs.append(cur); cur = []
    cur.append(k)
segs.append(cur)
seg = segs[-1]; t0 = seg[0][0]
for s, e, n, q in seg: print(f"{n:46s} {q:4s} {(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:.1f})")
PY
rm -rf "$OUT/trace"; grep call "$OUT/run.log"; cat "$OUT/net_gantt.txt"
