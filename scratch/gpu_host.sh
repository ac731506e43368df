#!/bin/bash
mkdir -p gpurun_out/host
echo skip > gpurun_out/host/tests.txt
: > gpurun_out/host/bench.txt
for v in "JM_AMD_DEC_OUT_FETCH=1/2" "JM_AMD_DEC_OUT_FETCH=1/3" "JM_AMD_DEC_OUT_FETCH=2/5" "JM_AMD_DEC_OUT_FETCH=3/5" "JM_AMD_DEC_OUT_FETCH=1/4" "JM_AMD_DEC_OUT_FETCH=1/2" "JM_AMD_DEC_OUT_FETCH=1/3"; do
  echo "== $v" >> gpurun_out/host/bench.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --steps 3 2>/dev/null | tail -1 >> gpurun_out/host/bench.txt
done
echo "== hevc" >> gpurun_out/host/bench.txt
timeout 300 python bench.py --codec hevc --streams 16 --frames 32 --width 1920 --height 1080 --steps 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/host/bench.txt
cat gpurun_out/host/tests.txt
python3 - <<'PY'
import json
for ln in open('gpurun_out/host/bench.txt'):
    if ln.startswith('=='): print(ln.strip()); continue
    try:
        j=json.loads(ln); h=j['host_cpu']; print(j['value'], h['cpus_busy'], h['cpu_ms_per_frame'], h['throttled_ms'], j['host_ms_per_picture'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in j['kernels'].items()})
    except Exception as e: print('bad', ln[:200])
PY
