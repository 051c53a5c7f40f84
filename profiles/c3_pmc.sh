#!/bin/bash
# Run ON THE GPU BOX from the repository root:  bash profiles/c3_pmc.sh <tag>  -- PMC passes (one counter set per run, kernel-trace only)
# over three fresh BarycentricInterpolator constructions of config 3 -> gpurun_out/<tag>/pmc_per_launch.json
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-c3pmc}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SET in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM"; do
    NAME=$(echo $SET | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_$NAME" -o pmc -- python $ROOT/profiles/c3_pmc_run.py > /dev/null 2> "$OUT/pmc_$NAME.log"
    find "$OUT/pmc_$NAME" -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_$NAME/pmc_counter_collection.csv" \; 2>/dev/null
done
cd $ROOT
python profiles/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +20M -delete
head -60 $OUT/summary.txt
