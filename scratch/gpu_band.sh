#!/bin/bash
mkdir -p gpurun_out/band
timeout 900 python -m pytest tests -m gpu -x -q -k "not hevc" 2>&1 | tail -8 > gpurun_out/band/tests.txt
: > gpurun_out/band/bench.txt
run() { echo "== $*" >> gpurun_out/band/bench.txt; timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> gpurun_out/band/bench.txt; }
run --steps 3
run --steps 3 --tools high
run --steps 3 --tools high_b
run --tools high_b --streams 32 --width 3840 --height 2160 --frames 12 --steps 2
run --steps 3 --streams 8
cat gpurun_out/band/tests.txt
python3 - <<'PY'
import json
for ln in open('gpurun_out/band/bench.txt'):
    if ln.startswith('=='): print(ln.strip()); continue
    try:
        j=json.loads(ln); print(' ', j['value'], 'err', j['decode_errors'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in j['kernels'].items()}, j['host_cpu']['cpus_busy'])
    except Exception as e: print('bad', ln[:200])
PY
