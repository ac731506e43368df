#!/bin/bash
# round 6: the early launches only with deep queues (hysteresis 16 / 8 pending pictures per decoder): headline and device-resident, 3 runs each, beside round 5's rule
mkdir -p gpurun_out/k; O=gpurun_out/k
python -m pytest tests/test_gpu_parity.py tests/test_pushpull.py -m gpu -x -q -k "32_streams or concurrent or c0_push or c4_slice" 2>&1 | tail -4 > $O/tests.log
for i in 1 2 3; do
  JM_AMD_DEC_CROSS_LANE=0 JM_AMD_DEC_EARLY_INTRA=0 python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_00_$i.json 2> $O/host_00_$i.err
  python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_11_$i.json 2> $O/host_11_$i.err
  JM_AMD_DEC_CROSS_LANE=0 JM_AMD_DEC_EARLY_INTRA=0 python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_00_$i.json 2> $O/dev_00_$i.err
  python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_11_$i.json 2> $O/dev_11_$i.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/k/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    e = d["engine"]; ln = e.get("lanes", {}); o = ln.get("ordinary", {}); it = ln.get("intra", {}); r = d["roofline"]
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], "cpus", d["host_cpu"]["cpus_busy"], "ord", o.get("pictures_per_batch"), o.get("busy_frac"), "intra", it.get("batches"), it.get("pictures_per_batch"), "early", ln.get("intra_pictures_launched_ahead_of_their_turn"), "roofline", r["kernel"], r["frac"], r["pictures_per_launch"])
PY
cat $O/tests.log
