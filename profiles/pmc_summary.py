import csv, sys, collections, glob, os, json
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    f = os.path.join(d, "pmc_counter_collection.csv")
    if not os.path.exists(f): continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xr::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg for c in agg[k]})
out = {k: {c: sum(v)/len(v) for c, v in d.items()} for k, d in agg.items()}
for k in sorted(out, key=lambda k: -out[k].get("SQ_WAVE_CYCLES", 0)):
    d = out[k]
    print(k)
    print("   ", "  ".join("%s=%.4g" % (c, d[c]) for c in names if c in d))
json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
