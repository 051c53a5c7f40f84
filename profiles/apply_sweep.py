"""`python profiles/apply_sweep.py [delaunay|lattice] VAR=VAL[,VAR=VAL...] ...` -- profiles/apply_k256_run.py once per environment
variant (plus "base"), one line each: ms per K = 256 apply and the plan kernel's own time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
kind = "delaunay"
if args and args[0] in ("delaunay", "lattice"):
    kind, args = args[0], args[1:]
for v in ["base"] + args:
    env = dict(os.environ)
    if v != "base":
        for kv in v.split(","):
            k, val = kv.split("=", 1)
            env[k] = val
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "apply_k256_run.py"), kind, "256", "10"], env=env, capture_output=True, text=True)
    try:
        r = json.loads(out.stdout.strip().splitlines()[-1])
        print("%-52s %.4f ms  %.1f %% of 8 TB/s   plan kernel %.4f ms per apply" % (v, r["ms_per_apply"], 100 * r["frac_of_8TBps"],
              r["kernel_ms"].get("apply_plan", [0, 0])[1] / 10), flush=True)
    except Exception:  # noqa: BLE001
        print(v, "FAILED", out.stderr[-300:], flush=True)
