#!/bin/bash
# round 5: chain launches, one reference window per WORKGROUP when its four macroblocks allow it (recon_device.h, "quad" path) -- parity first (chain tests, the
# H.264 parity file, a sweep: whole-stream mode runs chains), then HBM traffic of k_chain and the c4_slice-shaped / one-stream lines, head against the library
# before it (scratch/_ab/prev2)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/quad; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/gputests.log 2>&1; tail -2 $O/gputests.log
timeout 1500 python tools/gpu_sweep.py 150 70707 > $O/sweep.log 2>&1; tail -2 $O/sweep.log
for w in head prev2; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w != head ] && L=$GRAFT_REPO_ROOT/scratch/_ab/$w/libjm_amd_dec.so
  for s in 1 8; do for c in FETCH_SIZE WRITE_SIZE; do
    JM_AMD_DEC_LIB=$L rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_q -- python3 bench.py --steps 2 --warmup 1 --frames 30 --streams $s --no-cpu-baseline --no-single --no-extra --device-output > $O/line_${w}_${s}_$c.json 2>/dev/null
    f=$(find gpurun_out/prof_q -name "*counter_collection.csv" | head -1)
    python3 - "$f" $O/line_${w}_${s}_$c.json $w $s $c <<'PY'
import csv, json, sys, collections
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('jmamd::', '')
    tot[k] += float(r['Counter_Value']); n[k] += 1
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
frames = j['frames'] * 3 / 2     # (warm-up step included in the counters: 3 steps decoded, 2 timed)
for k in sorted(tot):
    if 'chain' in k: print(sys.argv[3], 'streams', sys.argv[4], sys.argv[5], k, 'dispatches', n[k], 'KB per picture of the run: %.0f' % (tot[k] / frames), 'bit_exact', j['bit_exact'])
PY
    rm -rf gpurun_out/prof_q
  done; done
done 2>&1 | tee $O/summary.txt
for w in head prev2 head prev2; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w != head ] && L=$GRAFT_REPO_ROOT/scratch/_ab/$w/libjm_amd_dec.so
  JM_AMD_DEC_LIB=$L timeout 300 python bench.py --streams 8 --frames 60 --steps 10 --no-extra --no-cpu-baseline > $O/c4_$w.json 2>/dev/null
  python3 - $O/c4_$w.json $w <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'c4_slice', j['value'], j['bit_exact'], {k: (v['avg_us'], v['pictures_per_launch']) for k, v in j['kernels'].items()}, 'recov', j['engine'].get('chain_recoveries_whole_run'), 'single', (j.get('single_stream') or {}).get('value'), 'devres', (j.get('device_resident_output') or {}).get('value'))
PY
done 2>&1 | tee -a $O/summary.txt
echo finished
