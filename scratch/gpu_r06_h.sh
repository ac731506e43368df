#!/bin/bash
# round 6: second soak (another box) with every hardware queue created at engine start, and the same shapes with lazy queue creation beside it:
# do the clock gaps (waves of a chain launch not run for 24-33 ms) come from queue creation?
mkdir -p gpurun_out/h; O=gpurun_out/h
python -m pytest tests/test_pushpull.py tests/test_hevc_gpu_parity.py -m gpu -x -q 2>&1 | tail -6 > $O/tests.log
P=gpurun_out/lazy; mkdir -p $P
for r in 1 2 3 4; do for s in 2 3 16; do
  JM_AMD_DEC_LAZY_QUEUES=1 JM_AMD_DEC_VERBOSE=1 timeout 600 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --streams $s > $P/lazy_s${s}_$r.json 2> $P/lazy_s${s}_$r.err
  JM_AMD_DEC_VERBOSE=1 timeout 600 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --streams $s > $P/eager_s${s}_$r.json 2> $P/eager_s${s}_$r.err
done; done
echo "lazy queue creation: runs with clock gaps $(grep -l 'clock gap' $P/lazy_*.err | wc -l) of $(ls $P/lazy_*.err | wc -l)"
echo "queues created at engine start: runs with clock gaps $(grep -l 'clock gap' $P/eager_*.err | wc -l) of $(ls $P/eager_*.err | wc -l)"
grep -h "clock gap" $P/*.err | cut -c1-260
cat $O/tests.log
bash scratch/gpu_soak_r06.sh > $O/soak.txt 2>&1
tail -16 $O/soak.txt | cut -c1-300
python bench.py > $O/bench_full.json 2> $O/bench_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/h/bench_full.json').read().strip().splitlines()[-1])
print(d['value'], d['bit_exact'], d['scaling_bound'], 'single', d.get('single_stream',{}).get('value'), 'c0', d.get('c0_pushpull',{}).get('value'), d.get('c0_pushpull',{}).get('vs_single_stream'), 'devres', d.get('device_resident_output',{}).get('value'), d.get('device_resident_output',{}).get('scaling_bound'))
for k in ("c4_slice","c2_4k","c3_4k"): print("   ", k, d[k].get("value"), d[k].get("scaling_bound"), d[k].get("bit_exact"), d[k].get("error"))
PY
