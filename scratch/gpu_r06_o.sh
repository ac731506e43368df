#!/bin/bash
# round 6: two copy streams for the job lists + the ordinary lane's fill linger.  c1l0 = round 6 so far (one copy stream, no linger); c2l0; c2l300; c2l600 (default); c2l900
mkdir -p gpurun_out/o; O=gpurun_out/o
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_list or concurrent or 32_streams or c4_slice or damaged" 2>&1 | tail -3 > $O/tests.log
for i in 1 2 3; do
  for cfg in c1l0 c2l0 c2l300 c2l600 c2l900; do
    c=${cfg:1:1}; l=${cfg#c?l}
    JM_AMD_DEC_COPY_STREAMS=$c JM_AMD_DEC_FILL_LINGER_US=$l python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_${cfg}_$i.json 2> $O/host_${cfg}_$i.err
    JM_AMD_DEC_COPY_STREAMS=$c JM_AMD_DEC_FILL_LINGER_US=$l python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_${cfg}_$i.json 2> $O/dev_${cfg}_$i.err
  done
done
for cfg in c1l0 c2l600; do
  c=${cfg:1:1}; l=${cfg#c?l}
  JM_AMD_DEC_COPY_STREAMS=$c JM_AMD_DEC_FILL_LINGER_US=$l python bench.py --streams 20 --no-extra --no-cpu-baseline --no-single > $O/s20_${cfg}_1.json 2> $O/s20_${cfg}_1.err
  JM_AMD_DEC_COPY_STREAMS=$c JM_AMD_DEC_FILL_LINGER_US=$l python bench.py --tools high_b --no-extra --no-cpu-baseline --no-single > $O/highb_${cfg}_1.json 2> $O/highb_${cfg}_1.err
  JM_AMD_DEC_COPY_STREAMS=$c JM_AMD_DEC_FILL_LINGER_US=$l python bench.py --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c2_${cfg}_1.json 2> $O/c2_${cfg}_1.err
  JM_AMD_DEC_COPY_STREAMS=$c JM_AMD_DEC_FILL_LINGER_US=$l python bench.py --codec hevc --width 3840 --height 2160 --streams 16 --frames 16 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c3_${cfg}_1.json 2> $O/c3_${cfg}_1.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/o/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    e = d["engine"]; ln = e.get("lanes", {}); o = ln.get("ordinary", ln.get("hevc", {})); r = d["roofline"]
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], "cpus", d["host_cpu"]["cpus_busy"], "ord", o.get("pictures_per_batch"), "busy", o.get("busy_frac"), "idle", o.get("idle_between_batches_frac"),
          "job lists", o.get("idle_waiting_for_job_lists_frac"), "dry", o.get("batches_launched_after_the_lane_ran_dry"), "of", o.get("batches"), "roofline", r["kernel"], r["frac"], r["pictures_per_launch"])
PY
cat $O/tests.log
