"""Average the rocprofv3 --pmc passes of profiles/collect.sh per kernel -> pmc_per_launch.json + pmc_traffic.json."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha():
    """sha1 over the kernel sources: bench.py refuses PMC numbers that belong to other kernels."""
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "xugrid_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "xugrid_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def short(name):
    return name.split("(")[0].replace("void ", "").replace("xr::", "").split("<")[0]


def main(root):
    # per (kernel, grid size): the same kernel launched with different grids (the persistent clip over the regular and
    # over the big pair queue) is kept apart -- the largest grid under the plain name, the others as name@grid
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        f = os.path.join(d, "pmc_counter_collection.csv")
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][int(r.get("Grid_Size", 0) or 0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, grids in agg.items():
        for g in grids:
            name = k if g == max(grids) else f"{k}@{g}"
            out[name] = {c: sum(v) / len(v) for c, v in grids[g].items()}
    names = sorted({c for k in out for c in out[k]})
    for k in sorted(out, key=lambda k: -out[k].get("SQ_WAVE_CYCLES", 0)):
        print(k)
        print("   ", "  ".join("%s=%.4g" % (c, out[k][c]) for c in names if c in out[k]))
    json.dump(out, open(os.path.join(root, "pmc_per_launch.json"), "w"), indent=1, sort_keys=True)
    traffic = {"_meta": {"source_sha": source_sha(), "unit": "bytes per launch",
                         "note": "hbm_bytes_corrected = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 reports half of a wide "
                                 "coalesced read, MI355X_MICROARCH.md section HBM; an upper bound for scattered access)"}}
    for k, d in out.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:  # rocprofv3 reports both in KB
            traffic[k] = {"fetch_bytes_raw": d["FETCH_SIZE"] * 1024, "write_bytes_raw": d["WRITE_SIZE"] * 1024,
                          "hbm_bytes_corrected": 2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024}
    json.dump(traffic, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
