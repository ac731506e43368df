#!/bin/bash
mkdir -p gpurun_out/h8
timeout 900 python -m pytest tests -m gpu -x -q -k "hevc" 2>&1 | tail -3 > gpurun_out/h8/tests.txt
: > gpurun_out/h8/bench.txt
echo "== hevc1080" >> gpurun_out/h8/bench.txt
timeout 300 python bench.py --codec hevc --streams 16 --frames 32 --width 1920 --height 1080 --steps 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/h8/bench.txt
echo "== hevc4k" >> gpurun_out/h8/bench.txt
timeout 300 python bench.py --codec hevc --streams 16 --frames 16 --width 3840 --height 2160 --steps 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/h8/bench.txt
cat gpurun_out/h8/tests.txt
python3 - <<'PY'
import json
for ln in open('gpurun_out/h8/bench.txt'):
    if ln.startswith('=='): print(ln.strip()); continue
    try:
        j=json.loads(ln); print(j['value'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in j['kernels'].items()}, j['roofline']['frac'], j['host_cpu']['cpus_busy'], j['host_cpu']['cpu_ms_per_frame'])
    except Exception as e: print('bad', ln[:200])
PY
