#!/bin/bash
# round 6: where the ordinary lane's idle time between batches goes (new lane diagnostics), and job-list uploads on 1 / 2 / 3 copy streams.  Device-resident and host-output.
mkdir -p gpurun_out/n; O=gpurun_out/n
for i in 1 2 3; do
  for c in 1 2 3; do
    JM_AMD_DEC_COPY_STREAMS=$c python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_c${c}_$i.json 2> $O/dev_c${c}_$i.err
  done
  for c in 1 2; do
    JM_AMD_DEC_COPY_STREAMS=$c python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_c${c}_$i.json 2> $O/host_c${c}_$i.err
  done
done
for c in 1 2; do
  JM_AMD_DEC_COPY_STREAMS=$c python bench.py --streams 1 --steps 20 --no-extra --no-cpu-baseline --no-single > $O/s1_c${c}_1.json 2> $O/s1_c${c}_1.err
  JM_AMD_DEC_COPY_STREAMS=$c python bench.py --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c2_c${c}_1.json 2> $O/c2_c${c}_1.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/n/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    e = d["engine"]; ln = e.get("lanes", {}); o = ln.get("ordinary", {}); r = d["roofline"]
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], "cpus", d["host_cpu"]["cpus_busy"], "ord", o.get("pictures_per_batch"), "busy", o.get("busy_frac"), "idle", o.get("idle_between_batches_frac"),
          "job lists", o.get("idle_waiting_for_job_lists_frac"), "pre-pass", o.get("idle_waiting_for_pre_pass_frac"), "dry", o.get("batches_launched_after_the_lane_ran_dry"), "of", o.get("batches"), "roofline", r["frac"])
PY
