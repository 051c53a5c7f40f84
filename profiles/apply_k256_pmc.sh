#!/bin/bash
# Run ON THE GPU BOX from the repository root:  bash profiles/apply_k256_pmc.sh <tag> [K]
# Memory-pipeline counters of the K-variable apply (BASELINE config 5) on the benchmark's qhull-numbered matrix and on the
# lattice-numbered pair, side by side: one rocprofv3 --pmc pass per counter set (kernel-trace only, never combined with other
# traces), plus a --kernel-trace --stats pass and a plain timed run of each.  -> gpurun_out/<tag>/{delaunay,lattice}/...
ROOT=$(pwd); TAG=${1:-applypmc}; K=${2:-256}
cd /tmp && export TMPDIR=/tmp
for KIND in delaunay lattice; do
    OUT=$ROOT/gpurun_out/$TAG/$KIND; mkdir -p $OUT
    python $ROOT/profiles/apply_k256_run.py $KIND $K 5 > $OUT/timed.json 2> $OUT/timed.log
    timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python $ROOT/profiles/apply_k256_run.py $KIND $K 5 > $OUT/under_rocprof.json 2> $OUT/stats.log
    cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
    rm -rf "$OUT/stats"
    # (small sets: a set that asks a hardware block for more counters than it has makes rocprofv3 abort and hang -- round 5 lost
    # 45 GPU-minutes to three such sets; every pass therefore runs under a short timeout)
    for SET in "FETCH_SIZE" "WRITE_SIZE" \
               "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" \
               "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" \
               "SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
               "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
               "TA_BUSY_avr TA_TA_BUSY_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
               "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
               "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
               "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
               "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
               "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum" "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" \
               "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_WRITE_sum TCC_TAG_STALL_sum" "TCC_BUSY_avr TCC_EA0_RDREQ_sum" \
               "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
               "TD_TD_BUSY_sum TD_TC_STALL_sum" "TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum"; do
        NAME=$(echo $SET | cut -d' ' -f1)
        timeout -k 5 150 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_$NAME" -o pmc -- python $ROOT/profiles/apply_k256_run.py $KIND $K 2 > /dev/null 2> "$OUT/pmc_$NAME.log"
        find "$OUT/pmc_$NAME" -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_$NAME/pmc_counter_collection.csv" \; 2>/dev/null
    done
    (cd $ROOT && python profiles/pmc_summary.py $OUT > $OUT/summary.txt 2>&1)
    find $OUT -name "*.csv" -size +8M -delete
    find $OUT -name "*.db" -delete
done
cd $ROOT
python profiles/apply_k256_table.py gpurun_out/$TAG > gpurun_out/$TAG/apply_plan_counters.txt 2>&1
cat gpurun_out/$TAG/apply_plan_counters.txt
