#!/bin/bash
mkdir -p gpurun_out/band
timeout 600 python -m pytest tests -m gpu -x -q -k "not hevc" 2>&1 | tail -4 > gpurun_out/band/tests.txt
: > gpurun_out/band/bench.txt
for v in "DEPTH=3" "DEPTH=2" "DEPTH=4" "DEPTH=3 PUB=1" "DEPTH=3 PUB=4"; do
  envs=""; for kv in $v; do envs="$envs JM_AMD_DEC_DEBLOCK_$kv"; done
  echo "== $v" >> gpurun_out/band/bench.txt
  env $envs timeout 300 python bench.py --no-cpu-baseline --steps 3 2>/dev/null | tail -1 >> gpurun_out/band/bench.txt
done
cat gpurun_out/band/tests.txt
python3 - <<'PY'
import json
for ln in open('gpurun_out/band/bench.txt'):
    if ln.startswith('=='): print(ln.strip()); continue
    try:
        j=json.loads(ln); print(j['value'], {k:v['avg_us'] for k,v in j['kernels'].items()}, j['roofline']['frac'], j['host_cpu']['cpus_busy'])
    except Exception as e: print('bad', ln[:200])
PY
