"""Side-by-side table of the apply kernels' counters on the qhull-numbered and the lattice-numbered matrix (profiles/apply_k256_pmc.sh)."""
import json
import os
import sys

root = sys.argv[1]
kinds = ["delaunay", "lattice"]
data = {}
timed = {}
for k in kinds:
    p = os.path.join(root, k, "pmc_per_launch.json")
    data[k] = json.load(open(p)) if os.path.exists(p) else {}
    try:
        timed[k] = json.loads(open(os.path.join(root, k, "timed.json")).read().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        timed[k] = {}
for k in kinds:
    t = timed[k]
    if t:
        print(f"{k:9s}: {t['ms_per_apply']:.4f} ms per apply, {t['algorithmic_GBps']:.0f} GB/s algorithmic = {100 * t['frac_of_8TBps']:.1f} % of 8 TB/s; "
              f"kernels (launches, ms total): { {n: (v[0], round(v[1], 4)) for n, v in t['kernel_ms'].items()} }")
kernels = sorted({n for k in kinds for n in data[k] if "apply" in n or "permute" in n})
for kern in kernels:
    print()
    print(kern)
    names = sorted({c for k in kinds for c in data[k].get(kern, {})})
    print("  %-44s %16s %16s %8s" % ("counter (per launch)", *kinds, "ratio"))
    for c in names:
        a = data["delaunay"].get(kern, {}).get(c)
        b = data["lattice"].get(kern, {}).get(c)
        fa = "%16.5g" % a if a is not None else " " * 16
        fb = "%16.5g" % b if b is not None else " " * 16
        r = "%8.2f" % (a / b) if a is not None and b else ""
        print("  %-44s %s %s %s" % (c, fa, fb, r))
