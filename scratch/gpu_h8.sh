#!/bin/bash
mkdir -p gpurun_out/h8
timeout 900 python -m pytest tests -m gpu -x -q -k "hevc" 2>&1 | tail -2 > gpurun_out/h8/tests.txt
: > gpurun_out/h8/bench.txt
for v in "A=1" "JM_AMD_DEC_EXP_HEVC=8" "A=1" "JM_AMD_DEC_EXP_HEVC=8"; do
  echo "== $v 1080p" >> gpurun_out/h8/bench.txt
  env $v timeout 300 python bench.py --codec hevc --streams 16 --frames 32 --width 1920 --height 1080 --steps 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/h8/bench.txt
done
for v in "A=1" "JM_AMD_DEC_EXP_HEVC=8"; do
  echo "== $v 4k" >> gpurun_out/h8/bench.txt
  env $v timeout 300 python bench.py --codec hevc --streams 16 --frames 16 --width 3840 --height 2160 --steps 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/h8/bench.txt
done
cat gpurun_out/h8/tests.txt
python3 - <<'PY'
import json
for ln in open('gpurun_out/h8/bench.txt'):
    if ln.startswith('=='): print(ln.strip()); continue
    try:
        j=json.loads(ln); print(' ', j['value'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in j['kernels'].items()})
    except Exception as e: print('bad', ln[:200])
PY
timeout 900 python tools/gpu_sweep.py 300 801 2>&1 | grep "^hevc 300\|MISMATCH\|FAIL" | cut -c1-200 | head -4
