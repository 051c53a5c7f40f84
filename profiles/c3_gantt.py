#!/usr/bin/env python
"""kernel_trace.csv of profiles/c3_gantt_run.py -> per construction (cut at pauses > 5 ms) the kernels with start / end (us) and queue."""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ker = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xr::", ""),
               r.get("Queue_Id", "0")) for r in rows), key=lambda x: x[0])
segs, cur = [], []
for k in ker:
    if cur and k[0] - max(c[1] for c in cur) > 5_000_000:
        segs.append(cur); cur = []
    cur.append(k)
segs.append(cur)
out = []
for seg in segs[-6:]:
    t0 = seg[0][0]
    busy, last, gaps = 0, t0, []
    for s, e, n, q in seg:
        if s > last: gaps.append((round((s - last) / 1e3, 1), n))
        if e > last: busy += e - max(s, last); last = e
    out.append({"wall_us": round((last - t0) / 1e3, 1), "busy_us": round(busy / 1e3, 1), "n": len(seg),
                "kernels": [f"{n[:28]:28s} q{q} {round((s - t0) / 1e3, 1):8.1f} -> {round((e - t0) / 1e3, 1):8.1f}" for s, e, n, q in seg],
                "gaps": sorted(gaps, reverse=True)[:8]})
print(json.dumps(out[2], indent=1)); print(json.dumps(out[5], indent=1))
