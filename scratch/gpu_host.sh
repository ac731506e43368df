#!/bin/bash
mkdir -p gpurun_out/host
: > gpurun_out/host/bench.txt
for v in "JM_AMD_DEC_THREADS=24"; do
  echo "== $v" >> gpurun_out/host/bench.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --steps 3 2>/dev/null | tail -1 >> gpurun_out/host/bench.txt
done
echo "== hevc" >> gpurun_out/host/bench.txt
JM_AMD_DEC_THREADS=24 timeout 300 python bench.py --codec hevc --streams 16 --frames 32 --width 1920 --height 1080 --steps 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/host/bench.txt
python3 - <<'PY'
import json
for ln in open('gpurun_out/host/bench.txt'):
    if ln.startswith('=='): print(ln.strip()); continue
    try:
        j=json.loads(ln); print(j['value'], j['host_cpu'], j['host_ms_per_picture'])
    except Exception as e: print('bad', ln[:200])
PY
