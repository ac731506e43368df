#!/bin/bash
mkdir -p gpurun_out/soak
: > gpurun_out/soak/bench.txt
run() { echo "== $*" >> gpurun_out/soak/bench.txt; timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> gpurun_out/soak/bench.txt; }
run --streams 64 --frames 300 --steps 1 --warmup 1
run --streams 8 --frames 300 --steps 2 --warmup 1
run --streams 16 --steps 3
run --streams 48 --steps 3
run --codec hevc --streams 32 --frames 32 --width 1920 --height 1080 --steps 2
run --codec hevc --streams 8 --frames 32 --width 1920 --height 1080 --steps 2
run --tools high_b --streams 32 --width 3840 --height 2160 --frames 12 --steps 2
python3 - <<'PY'
import json
for ln in open('gpurun_out/soak/bench.txt'):
    if ln.startswith('=='): print(ln.strip()); continue
    try:
        j=json.loads(ln); h=j['host_cpu']; print(' ', j['value'], 'errors', j['decode_errors'], 'frames', j['frames'], 'cpu', h['cpus_busy'], h['cpu_ms_per_frame'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in j['kernels'].items()})
    except Exception as e: print('bad', ln[:300])
PY
