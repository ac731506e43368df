#!/bin/bash
# round 6: soak (first box) + the quad-path test + lanes A/B with "early only when far from its turn"
mkdir -p gpurun_out/g; O=gpurun_out/g
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shared_reference_window or chain or recovered or damaged or bit_exact_vs_oracle or golden" 2>&1 | tail -8 > $O/tests.log
for i in 1 2; do
  JM_AMD_DEC_EARLY_INTRA=0 python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_10_$i.json 2> $O/dev_10_$i.err
  python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_11_$i.json 2> $O/dev_11_$i.err
  JM_AMD_DEC_EARLY_INTRA=0 python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_10_$i.json 2> $O/host_10_$i.err
  python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_11_$i.json 2> $O/host_11_$i.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/g/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    e = d["engine"]; ln = e.get("lanes", {}); o = ln.get("ordinary", {}); it = ln.get("intra", {})
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], "cpus", d["host_cpu"]["cpus_busy"], "ord", o.get("pictures_per_batch"), o.get("busy_frac"), "intra", it.get("batches"), it.get("pictures_per_batch"), "early", ln.get("intra_pictures_launched_ahead_of_their_turn"), "left-out ms", ln.get("left_out_ms_per_occasion"))
PY
cat $O/tests.log
bash scratch/gpu_soak_r06.sh > $O/soak.txt 2>&1
tail -50 $O/soak.txt
