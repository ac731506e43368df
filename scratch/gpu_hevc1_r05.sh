# round 5: HEVC device side -- parity tests on head, then head against the round-4 library: C3 4K / 1080p device-resident + kernel stats of a 4K run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; P=gpurun_out/hv1; mkdir -p $P
R4=$GRAFT_REPO_ROOT/scratch/_ab/r4/libjm_amd_dec.so
timeout 1200 python -m pytest tests/test_hevc_gpu_parity.py -m gpu -x -q > $P/gputests.log 2>&1; tail -3 $P/gputests.log
for i in 1 2; do
  for w in head r4; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = r4 ] && L=$R4
    JM_AMD_DEC_LIB=$L timeout 400 python bench.py --codec hevc --width 3840 --height 2160 --streams 16 --frames 16 --steps 3 --no-extra --no-cpu-baseline --no-single --device-output > $P/${w}_c34k_$i.json 2> $P/${w}_c34k_$i.err
    JM_AMD_DEC_LIB=$L timeout 400 python bench.py --codec hevc --streams 16 --frames 32 --steps 3 --no-extra --no-cpu-baseline --no-single --device-output > $P/${w}_c31080_$i.json 2> $P/${w}_c31080_$i.err
  done
done
python tools/ab_summary.py $P > $P/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/hv1/summary.json"))
for k,v in sorted(d.items()):
    print(k, v["value"], v["cpu_ms_per_frame"], {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in v["kernels"].items()})
PY
for w in head r4; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = r4 ] && L=$R4
  JM_AMD_DEC_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h -- python3 bench.py --codec hevc --width 3840 --height 2160 --streams 16 --frames 16 --steps 2 --no-extra --no-cpu-baseline --no-single --device-output > $P/${w}_under_rocprof.json 2>/dev/null
  find gpurun_out/prof_h -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $P/${w}_hevc_4k_kernel_stats.csv; rm -rf gpurun_out/prof_h
  head -8 $P/${w}_hevc_4k_kernel_stats.csv | cut -d, -f1-7
done
