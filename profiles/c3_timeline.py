#!/usr/bin/env python
"""kernel_trace.csv of profiles/c3_pmc_run.py (three fresh config-3 constructions) -> the LAST construction's kernels in launch
order with start / end relative to its first kernel, the union of busy time and the idle gaps (JSON on stdout)."""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ker = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xr::", ""),
               r.get("Queue_Id", "0")) for r in rows), key=lambda x: x[0])
# a construction starts with the Voronoi pre-step's first kernel (k_vor_init_counters)
starts = [i for i, k in enumerate(ker) if k[2].startswith("k_vor_init_counters")]
i0 = starts[-1]
# it may be preceded by the source-side kernels started first (centroids / locate_flag on the side stream): include kernels up to 400 us earlier
t_first = ker[i0][0]
j = i0
while j > 0 and t_first - ker[j - 1][1] < 400_000 and not ker[j - 1][2].startswith("k_bary_fill"):
    j -= 1
seg = ker[j:]
t0 = seg[0][0]
busy, last = 0, t0
gaps = []
for s, e, n, q in seg:
    if s > last:
        gaps.append((round((s - last) / 1e3, 1), n))
    if e > last:
        busy += e - max(s, last)
        last = e
out = {"wall_us": (last - t0) / 1e3, "busy_us": busy / 1e3, "idle_us": (last - t0 - busy) / 1e3,
       "kernels": [{"k": n[:32], "q": q, "start": round((s - t0) / 1e3, 1), "end": round((e - t0) / 1e3, 1)} for s, e, n, q in seg],
       "gaps_us_before": sorted(gaps, reverse=True)[:12]}
print(json.dumps(out, indent=1))
