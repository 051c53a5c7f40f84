#!/usr/bin/env python
"""kernel_trace.csv + hip_api_trace.csv of profiles/timeline_run.py -> per-step accounting (JSON on stdout):
where the wall time of a step goes that is NOT a kernel on the main chain -- idle gaps between dependent launches (with
the kernels on either side), what the host was blocked in meanwhile, the side stream's share."""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].split("<")[0].strip()
    return n.replace("void ", "").replace("xr::", "")


def med(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def main():
    kpath, hpath, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rows = list(csv.DictReader(open(kpath)))
    ker = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                   r.get("Queue_Id", "0"), r.get("Stream_Id", "0")) for r in rows), key=lambda x: x[0])
    # a step ends with the K = 1 apply kernel
    ends = [i for i, k in enumerate(ker) if k[2].startswith("k_apply_rows1")]
    ends = ends[-(steps + 1):]
    api = []
    if hpath:
        for r in csv.DictReader(open(hpath)):
            api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
        api.sort()
    main_queue = ker[ends[-1]][3]
    out_steps = []
    gap_by_pair = defaultdict(list)
    gantt = defaultdict(list)  # kernel name (+ occurrence within the step) -> [(start - t0, end - t0, on main queue)]
    for a, b in zip(ends[:-1], ends[1:]):
        t0, t1 = ker[a][1], ker[b][1]
        ks = ker[a + 1:b + 1]
        # union of all kernel intervals = device busy
        busy, cur_s, cur_e = 0, None, None
        gaps = []
        prev_name = ker[a][2]
        last_end = t0
        for s, e, n, q, st in sorted(ks, key=lambda x: x[0]):
            if s > last_end:
                gaps.append((s - last_end, prev_name, n, last_end, s))
            if e > last_end:
                busy += e - max(s, last_end)
                last_end = e
                prev_name = n
        seen = defaultdict(int)
        for s, e, n, q, st in ks:
            seen[n] += 1
            gantt[(n, seen[n])].append(((s - t0) / 1e3, (e - t0) / 1e3, q == main_queue))
        main_sum = sum(e - s for s, e, n, q, st in ks if q == main_queue)
        side_sum = sum(e - s for s, e, n, q, st in ks if q != main_queue)
        # what the host was inside during each idle gap
        gap_rows = []
        for g, pn, nn, gs, ge in sorted(gaps, reverse=True)[:40]:
            inside = defaultdict(int)
            for s, e, f in api:
                if e <= gs or s >= ge:
                    continue
                inside[f] += min(e, ge) - max(s, gs)
            top = sorted(inside.items(), key=lambda kv: -kv[1])[:3]
            gap_rows.append({"us": g / 1e3, "after": pn, "before": nn, "host_in": {k: round(v / 1e3, 2) for k, v in top}})
            gap_by_pair[(pn, nn)].append(g / 1e3)
        host_block = defaultdict(int)
        for s, e, f in api:
            if e <= t0 or s >= t1:
                continue
            if "Synchronize" in f or "hipMemcpy" in f:
                host_block[f] += min(e, t1) - max(s, t0)
        n_launch = sum(1 for s, e, f in api if t0 <= s < t1 and "LaunchKernel" in f)
        out_steps.append({
            "wall_us": (t1 - t0) / 1e3, "device_busy_us": busy / 1e3, "idle_us": (t1 - t0 - busy) / 1e3,
            "kernels": len(ks), "launch_calls": n_launch, "main_queue_kernel_sum_us": main_sum / 1e3,
            "side_queue_kernel_sum_us": side_sum / 1e3, "n_gaps": len(gaps),
            "host_blocked_us": {k: round(v / 1e3, 2) for k, v in host_block.items()},
            "largest_gaps": gap_rows[:12],
        })
    n = max(1, len(out_steps))
    pairs = sorted(((k, sum(v) / n) for k, v in gap_by_pair.items()), key=lambda kv: -kv[1])
    summary = {
        "what": "rocprofv3 --hip-trace --kernel-trace of profiles/timeline_run.py: per step (end of one K=1 apply kernel to the "
                "end of the next) wall, time with at least one kernel running on any queue, idle time, and the idle gaps by the "
                "kernels on either side (mean us per step); `gantt_us` and `median`: medians over the steps (a step that catches a host hiccup moves a mean by tens of microseconds)",
        "steps": len(out_steps),
        "mean": {k: sum(s[k] for s in out_steps) / n for k in ("wall_us", "device_busy_us", "idle_us", "kernels", "launch_calls",
                                                               "main_queue_kernel_sum_us", "side_queue_kernel_sum_us", "n_gaps")},
        "idle_by_gap_us_per_step": [{"after": a, "before": b, "us": round(v, 2)} for (a, b), v in pairs[:30]],
        "gantt_us": [{"kernel": k[0] + ("" if k[1] == 1 else "#%d" % k[1]), "queue": "main" if v[0][2] else "side",
                      "start": round(med([x[0] for x in v]), 2), "end": round(med([x[1] for x in v]), 2)}
                     for k, v in sorted(gantt.items(), key=lambda kv: med([x[0] for x in kv[1]]))],
        "median": {k: med([s[k] for s in out_steps]) for k in ("wall_us", "device_busy_us", "idle_us", "main_queue_kernel_sum_us",
                                                               "side_queue_kernel_sum_us")},
        "per_step": out_steps,
    }
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
