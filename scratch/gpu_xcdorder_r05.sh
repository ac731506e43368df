#!/bin/bash
# round 5: chain work list, groups of one key ordered by XCD (engine.cpp) -- HBM traffic of k_chain (FETCH_SIZE / WRITE_SIZE per picture), head against the
# previous library (scratch/_ab/prev), one stream and eight streams; then the chain parity tests and the c4_slice-shaped line for the kernel time
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/xcd; mkdir -p $O
for w in head prev; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = prev ] && L=$GRAFT_REPO_ROOT/scratch/_ab/prev/libjm_amd_dec.so
  for s in 1 8; do for c in FETCH_SIZE WRITE_SIZE; do
    JM_AMD_DEC_LIB=$L rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_x -- python3 bench.py --steps 2 --warmup 1 --frames 30 --streams $s --no-cpu-baseline --no-single --no-extra --device-output > $O/line_${w}_${s}_$c.json 2>/dev/null
    f=$(find gpurun_out/prof_x -name "*counter_collection.csv" | head -1)
    python3 - "$f" $O/line_${w}_${s}_$c.json $w $s $c <<'PY'
import csv, json, sys, collections
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('jmamd::', '')
    tot[k] += float(r['Counter_Value']); n[k] += 1
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
frames = j['frames'] * 3 / 2     # (warm-up step included in the counters: 3 steps decoded, 2 timed)
for k in tot:
    if 'chain' in k: print(sys.argv[3], 'streams', sys.argv[4], sys.argv[5], k, 'dispatches', n[k], 'per picture: %.0f (unit as counted: 32 B for FETCH_SIZE? KB for WRITE_SIZE)' % (tot[k] / frames), 'sum', tot[k])
PY
    rm -rf gpurun_out/prof_x
  done; done
done 2>&1 | tee $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain" > $O/gputests.log 2>&1; tail -2 $O/gputests.log
for w in head prev head prev; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = prev ] && L=$GRAFT_REPO_ROOT/scratch/_ab/prev/libjm_amd_dec.so
  JM_AMD_DEC_LIB=$L timeout 300 python bench.py --streams 8 --frames 60 --steps 10 --no-extra --no-cpu-baseline --no-single > $O/c4_$w.json 2>/dev/null
  python3 - $O/c4_$w.json $w <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'c4_slice', j['value'], j['bit_exact'], {k: (v['avg_us'], v['pictures_per_launch']) for k, v in j['kernels'].items()}, j['engine'].get('chain_recoveries_whole_run'))
PY
done
echo finished
