#!/bin/bash
# round 6: C3 after HEVC uploads went back to copy stream 0 (default build) beside JM_AMD_DEC_COPY_STREAMS=1; C2 alike; HEVC GPU tests
mkdir -p gpurun_out/u; O=gpurun_out/u
for i in 1 2 3; do
  for c in 1 2; do
    JM_AMD_DEC_COPY_STREAMS=$c python bench.py --codec hevc --width 3840 --height 2160 --streams 16 --frames 16 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c3_c${c}_$i.json 2> $O/c3_c${c}_$i.err
    JM_AMD_DEC_COPY_STREAMS=$c python bench.py --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c2_c${c}_$i.json 2> $O/c2_c${c}_$i.err
  done
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/u/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], d.get("bound_utilisation"), "cpu ms/frame", d["host_cpu"]["cpu_ms_per_frame"])
PY
timeout 900 python -m pytest tests/test_hevc_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
