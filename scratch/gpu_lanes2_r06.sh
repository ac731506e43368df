#!/bin/bash
# round 6: (a) time line of the lanes with and without intra pictures launched ahead (why was it slower?), (b) the chain tests and a third of the soak
# with the new wait timers (chain_common.h WaitClock)
mkdir -p gpurun_out/soak
for cfg in "0 0" "1 1"; do set -- $cfg
  JM_AMD_DEC_LANE_TRACE=6000 JM_AMD_DEC_CROSS_LANE=$1 JM_AMD_DEC_EARLY_INTRA=$2 python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 2 --warmup 1 > gpurun_out/r06_trace_$1$2.json 2> gpurun_out/r06_trace_$1$2.err
done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain or recovered or damaged or zz" 2>&1 | tail -8 > gpurun_out/r06_chain_tests.log
P=gpurun_out/soak
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs JM_AMD_DEC_VERBOSE=1 timeout 600 python bench.py --no-extra --no-cpu-baseline --no-single "$@" > $P/$tag.json 2> $P/$tag.err; }
for s in 1 2 4 6 8 16; do run c1_s${s}_r1 -- --steps 20 --streams $s; done
run c2_chain_1 JM_AMD_DEC_CHAIN_STREAMS=64 -- --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3
run highb_s4_1 -- --tools high_b --streams 4 --steps 10
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/soak/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(os.path.basename(f), 'NO LINE'); continue
    e = d["engine"]
    print(os.path.basename(f), d["value"], "bit_exact", d["bit_exact"], "chain launches", e["chain_batches_whole_run"], "recoveries", e["chain_recoveries_whole_run"], "wait errors", e["device_wait_errors"], "gaps", e.get("chain_launches_with_clock_gaps"), e.get("longest_clock_gap_us_whole_process"))
PY
grep -l "gave up\|clock gap" $P/*.err | head
cat gpurun_out/r06_chain_tests.log
grep -c lane-trace gpurun_out/r06_trace_*.err
