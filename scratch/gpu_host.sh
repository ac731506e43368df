#!/bin/bash
mkdir -p gpurun_out/host
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/host/tests.txt
: > gpurun_out/host/bench.txt
for v in "A=1" "JM_AMD_DEC_PLAIN_COPY=1" "A=1" "JM_AMD_DEC_PLAIN_COPY=1" "JM_AMD_DEC_THREADS=20" "JM_AMD_DEC_THREADS=28"; do
  echo "== $v" >> gpurun_out/host/bench.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --steps 3 2>/dev/null | tail -1 >> gpurun_out/host/bench.txt
done
cat gpurun_out/host/tests.txt
python3 - <<'PY'
import json
for ln in open('gpurun_out/host/bench.txt'):
    if ln.startswith('=='): print(ln.strip()); continue
    try:
        j=json.loads(ln); h=j['host_cpu']; print(j['value'], h['cpus_busy'], h['cpu_ms_per_frame'], h['throttled_ms'], h['by_thread'].get('jm-parse'), j['host_ms_per_picture'])
    except Exception as e: print('bad', ln[:200])
PY
