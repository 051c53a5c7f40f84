#!/bin/bash
# Run ON THE GPU BOX from the repository root:  bash profiles/fma_ab.sh <tag>
# The price of bit-exactness: the search + clip translation unit built with -ffp-contract=fast (make -C xugrid_amd/csrc fma ->
# libxugrid_amd_fma.so, an A/B artefact the product never loads) against the default -ffp-contract=off build: pair set and areas
# against the CPU oracle at full size (1M -> 1M), kernel times (interleaved, 3 rounds), VALU instruction counts (one PMC pass each).
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-fma_ab}; mkdir -p $OUT
FMA=$ROOT/xugrid_amd/libxugrid_amd_fma.so
[ -f $FMA ] || make -C xugrid_amd/csrc fma > $OUT/make.log 2>&1
python profiles/fma_ab_run.py 50 > $OUT/off_0.json 2> $OUT/off.log
XUGRID_AMD_LIB=$FMA python profiles/fma_ab_run.py 50 > $OUT/fma_0.json 2> $OUT/fma.log
for r in 1 2; do
    python profiles/fma_ab_run.py 50 --no-oracle > $OUT/off_$r.json 2>> $OUT/off.log
    XUGRID_AMD_LIB=$FMA python profiles/fma_ab_run.py 50 --no-oracle > $OUT/fma_$r.json 2>> $OUT/fma.log
done
cd /tmp && export TMPDIR=/tmp
for V in off fma; do
    [ $V = fma ] && export XUGRID_AMD_LIB=$FMA || unset XUGRID_AMD_LIB
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_$V -o pmc -- python $ROOT/profiles/fma_ab_run.py 3 --no-oracle > /dev/null 2> $OUT/pmc_$V.log
    find $OUT/pmc_$V -name '*counter_collection.csv' -exec cp {} $OUT/pmc_$V.csv \; 2>/dev/null
    rm -rf $OUT/pmc_$V
done
unset XUGRID_AMD_LIB
cd $ROOT
python - $OUT <<'PY' | tee $OUT/fma_ab.txt
import collections, csv, json, sys, os
out = sys.argv[1]
def med(xs): return sorted(xs)[len(xs) // 2]
res = {}
for v in ("off", "fma"):
    runs = [json.loads(open(os.path.join(out, f"{v}_{r}.json")).read().strip().splitlines()[-1]) for r in range(3)]
    res[v] = runs
    valu = collections.defaultdict(list)
    p = os.path.join(out, f"pmc_{v}.csv")
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == "SQ_INSTS_VALU":
                valu[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xr::", "")].append(float(r["Counter_Value"]))
    print(f"== -ffp-contract={'fast' if v == 'fma' else 'off'} ({runs[0]['lib']})")
    print("   weights build ms (median of 3 x 50 steps):", round(med([r["weights_ms"] for r in runs]), 4))
    for k in runs[0]["kernel_ms_per_step"]:
        print(f"   {k:12s} ms:", round(med([r["kernel_ms_per_step"][k] for r in runs]), 5))
    for k, xs in sorted(valu.items()):
        if any(t in k for t in ("k_search", "k_clip_tri", "k_assemble")):
            print(f"   SQ_INSTS_VALU per launch {k[:60]:60s}: {sum(xs) / len(xs) / 1e6:.2f} M ({len(xs)} launches)")
    print("   parity vs oracle at 1M -> 1M:", json.dumps({k: runs[0][k] for k in runs[0] if k not in ("kernel_ms_per_step", "lib", "weights_ms", "weights_ms_profiled")}))
PY
