#!/bin/bash
# Developer script (gpurun): the default configuration (C1, 32 streams) against the size of the parse pool, two runs each
mkdir -p gpurun_out
for rep in a b; do for t in 12 14 16 18 20 24; do
  JM_AMD_DEC_THREADS=$t python bench.py --steps 3 --no-cpu-baseline --no-single > gpurun_out/thr3_base_${t}_$rep.json 2> gpurun_out/thr3_base_${t}_$rep.err
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/thr3_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['host_cpu']['cpu_ms_per_frame'], d['host_cpu']['cpus_busy'], d['host_cpu']['throttled_ms'], d['scaling_bound'], d['host_ms_per_picture'])
    except Exception as e: print(f, 'failed', e)
PY
