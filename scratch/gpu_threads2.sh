#!/bin/bash
# Developer script (gpurun): the other configurations against the size of the parse pool
mkdir -p gpurun_out
for t in 16 20 28; do
  JM_AMD_DEC_THREADS=$t python bench.py --steps 3 --no-cpu-baseline --no-single > gpurun_out/thr_base_$t.json 2> gpurun_out/thr_base_$t.err
  JM_AMD_DEC_THREADS=$t python bench.py --tools high_b --steps 3 --no-cpu-baseline --no-single > gpurun_out/thr_highb_$t.json 2> gpurun_out/thr_highb_$t.err
  JM_AMD_DEC_THREADS=$t python bench.py --tools high_b --width 3840 --height 2160 --frames 16 --steps 3 --no-cpu-baseline --no-single > gpurun_out/thr_c2_$t.json 2> gpurun_out/thr_c2_$t.err
  JM_AMD_DEC_THREADS=$t python bench.py --codec hevc --streams 16 --frames 16 --width 3840 --height 2160 --steps 3 --no-cpu-baseline --no-single > gpurun_out/thr_hevc4k_$t.json 2> gpurun_out/thr_hevc4k_$t.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/thr_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['host_cpu']['cpu_ms_per_frame'], d['host_cpu']['cpus_busy'], d['host_cpu']['throttled_ms'], d['host_ms_per_picture'])
    except Exception as e: print(f, 'failed', e)
PY
