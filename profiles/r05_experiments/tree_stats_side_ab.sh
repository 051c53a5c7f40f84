timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for e in 0 1 0 1; do echo == XR_TREE_STATS_SIDE=$e; XR_TREE_STATS_SIDE=$e timeout 300 python bench.py --no-cpu --step-only --no-extras 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], b['config']['weights_only_ms'])"; done
