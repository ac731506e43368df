#!/bin/bash
# round 6: device-resident leg, the final rules beside round 5's, eight alternating runs each (is the 22.3 k run of scratch/gpu_r06_k.sh the box or the rule?)
mkdir -p gpurun_out/l; O=gpurun_out/l
for i in 1 2 3 4 5 6 7 8; do
  JM_AMD_DEC_CROSS_LANE=0 JM_AMD_DEC_EARLY_INTRA=0 python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_00_$i.json 2> $O/dev_00_$i.err
  python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_11_$i.json 2> $O/dev_11_$i.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/l/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    e = d["engine"]; ln = e.get("lanes", {}); o = ln.get("ordinary", {}); it = ln.get("intra", {}); r = d["roofline"]
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], "cpus", d["host_cpu"]["cpus_busy"], "ord", o.get("pictures_per_batch"), o.get("busy_frac"), "intra", it.get("batches"), it.get("pictures_per_batch"), "early", ln.get("intra_pictures_launched_ahead_of_their_turn"), "roofline", r["frac"], r["pictures_per_launch"])
PY
