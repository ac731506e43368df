#!/bin/bash
# round 6: the fill linger with the urgency rule (a picture whose handle has no decoded, unfetched frame left goes at once): 0 / 2000 (default) / 4000 us.  Two copy streams.
mkdir -p gpurun_out/r; O=gpurun_out/r
for i in 1 2 3; do
  for l in 0 2000 4000; do
    JM_AMD_DEC_FILL_LINGER_US=$l python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_l${l}_$i.json 2> $O/host_l${l}_$i.err
  done
  for l in 0 2000; do
    JM_AMD_DEC_FILL_LINGER_US=$l python bench.py --streams 20 --no-extra --no-cpu-baseline --no-single > $O/s20_l${l}_$i.json 2> $O/s20_l${l}_$i.err
    JM_AMD_DEC_FILL_LINGER_US=$l python bench.py --tools high_b --no-extra --no-cpu-baseline --no-single > $O/highb_l${l}_$i.json 2> $O/highb_l${l}_$i.err
    JM_AMD_DEC_FILL_LINGER_US=$l python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_l${l}_$i.json 2> $O/dev_l${l}_$i.err
  done
done
for l in 0 2000; do
  JM_AMD_DEC_FILL_LINGER_US=$l python bench.py --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c2_l${l}_1.json 2> $O/c2_l${l}_1.err
  JM_AMD_DEC_FILL_LINGER_US=$l python bench.py --tools high --no-extra --no-cpu-baseline --no-single > $O/high_l${l}_1.json 2> $O/high_l${l}_1.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    e = d["engine"]; ln = e.get("lanes", {}); o = ln.get("ordinary", {}); it = ln.get("intra", {}); r = d["roofline"]
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], "cpus", d["host_cpu"]["cpus_busy"], "ord", o.get("pictures_per_batch"), "busy", o.get("busy_frac"), "idle", o.get("idle_between_batches_frac"),
          "dry", o.get("batches_launched_after_the_lane_ran_dry"), "of", o.get("batches"), "intra", it.get("batches"), it.get("pictures_per_batch"), "roofline", r["kernel"], r["frac"], r["pictures_per_launch"], "pcie", (d.get("pcie_out") or {}).get("frac"))
PY
