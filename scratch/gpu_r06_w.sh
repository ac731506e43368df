#!/bin/bash
# round 6: second copy stream created AFTER the lanes' streams: C3 / C2 with 1 and 2 copy streams, alternating; device-resident and headline once each way
mkdir -p gpurun_out/w; O=gpurun_out/w
for i in 1 2 3; do
  for c in 1 2; do
    JM_AMD_DEC_COPY_STREAMS=$c python bench.py --codec hevc --width 3840 --height 2160 --streams 16 --frames 16 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c3_c${c}_$i.json 2> $O/c3_c${c}_$i.err
    JM_AMD_DEC_COPY_STREAMS=$c python bench.py --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c2_c${c}_$i.json 2> $O/c2_c${c}_$i.err
  done
done
for i in 1 2; do
  for c in 1 2; do
    JM_AMD_DEC_COPY_STREAMS=$c python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_c${c}_$i.json 2> $O/dev_c${c}_$i.err
    JM_AMD_DEC_COPY_STREAMS=$c python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_c${c}_$i.json 2> $O/host_c${c}_$i.err
  done
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/w/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    ln = d["engine"].get("lanes", {}); o = ln.get("ordinary", ln.get("hevc", {}))
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], d.get("bound_utilisation"), "cpu ms/frame", d["host_cpu"]["cpu_ms_per_frame"], "lane", o.get("pictures_per_batch"), o.get("busy_frac"), "job lists", o.get("idle_waiting_for_job_lists_frac"), "roofline", d["roofline"]["frac"])
PY
