#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root:   bash profiles/collect.sh <tag> [bench args...]
# Produces under gpurun_out/<tag>/ :
#   kernel_stats.csv     rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu --no-extras --steps 10`
#   bench_under_rocprof.json   the bench line of that run
#   pmc_<set>/           one rocprofv3 --pmc pass per counter set (kernel-trace only; never combined with other traces)
#   pmc_per_launch.json  per-kernel averages of all passes (profiles/pmc_summary.py)
#   pmc_traffic.json     HBM-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction of
#                        MI355X_MICROARCH.md section HBM) + the sha of the kernel sources the numbers belong to
set -u
TAG=${1:-prof}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu --no-extras --step-only $*"
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH --steps 10 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
for SET in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM"; do
    NAME=$(echo $SET | cut -d' ' -f1)
    timeout -k 5 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_$NAME" -o pmc -- $BENCH --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_$NAME.log"
    # keep only the counter csv (the directories hold one file per process)
    find "$OUT/pmc_$NAME" -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_$NAME/pmc_counter_collection.csv" \; 2>/dev/null
done
cd "$ROOT"
python profiles/pmc_summary.py "$OUT"
