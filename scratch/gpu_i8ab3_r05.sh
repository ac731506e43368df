#!/bin/bash
# round 5: Intra8x8 blocks unrolled (availability folds for the inner blocks) + HEVC levels scaled on the device: probe, both parity suites, a sweep, then C3 4K
# head against the previous library (scratch/_ab/prev) on this box
cd "$GRAFT_REPO_ROOT" || exit 1
bash scratch/gpu_i8probe_r05.sh 2>&1 | grep -E "^==|k_intra_band"
O=gpurun_out/i8; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_hevc_gpu_parity.py -m gpu -x -q > $O/gputests3.log 2>&1; tail -3 $O/gputests3.log
timeout 1500 python tools/gpu_sweep.py 120 60606 > $O/sweep3.log 2>&1; tail -3 $O/sweep3.log
for rep in 1 2; do for w in head prev; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = prev ] && L=$GRAFT_REPO_ROOT/scratch/_ab/prev/libjm_amd_dec.so
  JM_AMD_DEC_LIB=$L timeout 600 python bench.py --codec hevc --width 3840 --height 2160 --streams 16 --frames 16 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c3_${w}_$rep.json 2>/dev/null
  python - $O/c3_${w}_$rep.json $w <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'c3_4k', j['value'], j.get('bit_exact'), j.get('scaling_bound'), j['host_cpu']['cpu_ms_per_frame'], {k: (v['avg_us'], v['pictures_per_launch']) for k, v in j.get('kernels', {}).items()})
PY
done; done
echo finished
