#!/usr/bin/env python
"""A/B on ONE box: `python profiles/ab.py [--steps N] [--reps R] NAME=VALUE[,NAME=VALUE...] ...` runs bench.py --no-cpu --no-extras once per
variant ("base" = no variable set) -- interleaved, R rounds -- and prints step / kernel times per variant (median over rounds)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
steps, reps = 100, 3
while args and args[0].startswith("--"):
    if args[0] == "--steps":
        steps = int(args[1])
    elif args[0] == "--reps":
        reps = int(args[1])
    args = args[2:]
variants = ["base"] + args
res = {v: [] for v in variants}
for _ in range(reps):
    for v in variants:
        env = dict(os.environ)
        if v != "base":
            for kv in v.split(","):
                k, val = kv.split("=", 1)
                env[k] = val
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-extras", "--steps", str(steps), "--warmup", "20"],
                             env=env, capture_output=True, text=True)
        try:
            res[v].append(json.loads(out.stdout.strip().splitlines()[-1]))
        except Exception:
            print(v, "FAILED", out.stderr[-400:])
keys = ["search", "place_big", "clip_tri", "assemble", "search_big", "clip_big", "row_fill_long", "row_fill_huge", "index_scatter", "index_count", "prepare_faces", "apply_rows1"]
for v in variants:
    if not res[v]:
        continue
    med = lambda xs: sorted(xs)[len(xs) // 2]
    line = {"ms_per_step": round(med([r["ms_per_step"] for r in res[v]]), 4),
            "weights_only": round(med([r["config"]["weights_only_ms"] for r in res[v]]), 4),
            "apply_only": round(med([r["config"]["apply_only_ms"] for r in res[v]]), 4)}
    for k in keys:
        xs = [r["roofline"]["kernel_ms_per_step"].get(k) for r in res[v]]
        xs = [x for x in xs if x is not None]
        if xs:
            line[k] = round(med(xs), 4)
    print(v, json.dumps(line))
