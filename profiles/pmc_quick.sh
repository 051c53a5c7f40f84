#!/bin/bash
# bash profiles/pmc_quick.sh <tag> "<counter set>;<counter set>;..." <python script + args...>   (environment passes through)
# One rocprofv3 --pmc pass per set (kernel-trace only, short timeout each) -> gpurun_out/<tag>/pmc_per_launch.json + summary.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; SETS=$2; shift 2
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra ARR <<< "$SETS"
for SET in "${ARR[@]}"; do
    NAME=$(echo $SET | cut -d' ' -f1)
    timeout -k 5 150 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_$NAME" -o pmc -- python "$ROOT/$1" "${@:2}" > /dev/null 2> "$OUT/pmc_$NAME.log"
    find "$OUT/pmc_$NAME" -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_$NAME/pmc_counter_collection.csv" \; 2>/dev/null
done
cd $ROOT && python profiles/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +8M -delete; find $OUT -name "*.db" -delete
