#!/bin/bash
# round 5: quick check of a tree after a cosmetic source change -- the H.264 parity file (chain launches included) and an eight-stream line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/q5
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/q5/gputests.log 2>&1; tail -2 gpurun_out/q5/gputests.log
timeout 300 python bench.py --streams 8 --no-extra --no-cpu-baseline --no-single > gpurun_out/q5/s8.json 2>/dev/null
python3 - <<'PY'
import json
j = json.loads(open('gpurun_out/q5/s8.json').read().strip().splitlines()[-1])
print('s8', j['value'], j['bit_exact'], j['engine']['chain_recoveries_whole_run'], {k: (v['avg_us'], v['pictures_per_launch']) for k, v in j['kernels'].items()})
PY
echo finished
