#!/bin/bash
# round 6: copy streams with a stream priority of their own (separate hardware-queue pool): JM_AMD_DEC_COPY_PRIORITY = 0 (the lanes' own: order 'beside', as in the
# first final set) / 1 (high, the default) / -1 (low), every leg, alternating
mkdir -p gpurun_out/x; O=gpurun_out/x
for i in 1 2; do
  for p in 0 1 -1; do
    export JM_AMD_DEC_COPY_PRIORITY=$p
    python bench.py --streams 8 --no-extra --no-cpu-baseline --no-single > $O/s8_p${p}_$i.json 2> $O/s8_p${p}_$i.err
    python bench.py --streams 1 --steps 20 --no-extra --no-cpu-baseline --no-single > $O/s1_p${p}_$i.json 2> $O/s1_p${p}_$i.err
    python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_p${p}_$i.json 2> $O/dev_p${p}_$i.err
    python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_p${p}_$i.json 2> $O/host_p${p}_$i.err
    python bench.py --codec hevc --width 3840 --height 2160 --streams 16 --frames 16 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c3_p${p}_$i.json 2> $O/c3_p${p}_$i.err
    python bench.py --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c2_p${p}_$i.json 2> $O/c2_p${p}_$i.err
  done
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/x/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    ln = d["engine"].get("lanes", {}); o = ln.get("ordinary", ln.get("hevc", {}))
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], "cpu ms/frame", d["host_cpu"]["cpu_ms_per_frame"], "lane", o.get("pictures_per_batch"), o.get("busy_frac"), "job lists", o.get("idle_waiting_for_job_lists_frac"), "roofline", d["roofline"]["frac"])
PY
