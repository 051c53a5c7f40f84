timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for e in 0 16; do echo == XR_BIG_EARLY=$e; XR_BIG_EARLY=$e timeout 300 python bench.py --no-cpu --step-only --no-extras 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], b['config']['weights_only_ms'])"; done
bash profiles/timeline.sh tl7 10 > /dev/null 2>&1
python - <<PY
import json
t=json.load(open("gpurun_out/tl7/timeline_summary.json"))
for g in t["gantt_us"]: print(g["kernel"][:22], g["queue"], g["start"], g["end"])
print(t["median"])
PY
