#!/bin/bash
# round 6, VERDICT item 2: why ordinary-lane batches are not full, and what the two changes of Engine::form buy (one box, one call)
mkdir -p gpurun_out
python -m pytest tests/test_pushpull.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06_pushpull_tests.log
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "concurrent or c4_slice or full_size or b_streams or mixed_res or display_delay or resolution_change" 2>&1 | tail -8 > gpurun_out/r06_parity_subset.log
for cfg in "0 0" "1 0" "1 1"; do
  set -- $cfg
  JM_AMD_DEC_CROSS_LANE=$1 JM_AMD_DEC_EARLY_INTRA=$2 python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > gpurun_out/r06_lanes_dev_$1$2.json 2> gpurun_out/r06_lanes_dev_$1$2.err
done
JM_AMD_DEC_CROSS_LANE=0 JM_AMD_DEC_EARLY_INTRA=0 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r06_lanes_host_00.json 2> gpurun_out/r06_lanes_host_00.err
python bench.py --no-extra --no-cpu-baseline > gpurun_out/r06_lanes_host_11.json 2> gpurun_out/r06_lanes_host_11.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_lanes_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    print(f, d['value'], d['bit_exact'], d['scaling_bound'], json.dumps(d['engine']['lanes']), d['host_cpu']['cpus_busy'], d.get('single_stream',{}).get('value'), d.get('c0_pushpull'), d.get('device_resident_output',{}).get('value'))
PY
cat gpurun_out/r06_pushpull_tests.log gpurun_out/r06_parity_subset.log
