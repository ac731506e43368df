# round 4: the band above's step counter sampled through the prefetch stage (JM_DEBLOCK_ASYNC_POLL=1, 'apoll') against head: suite on apoll, then device-resident
# default workload, 1 / 8 streams, C2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab8
AP=$GRAFT_REPO_ROOT/jmcodec_amd/lib_dbg_apoll/libjm_amd_dec.so
JM_AMD_DEC_LIB=$AP timeout 1500 python -m pytest tests -m gpu -x -q -rs > gpurun_out/ab8/gputests_apoll.log 2>&1; tail -4 gpurun_out/ab8/gputests_apoll.log
python bench.py --no-extra --no-cpu-baseline --no-single --steps 3 > /dev/null 2>&1
for i in 1 2 3; do
  for w in head apoll; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = apoll ] && L=$AP
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --device-output > gpurun_out/ab8/${w}_dev_$i.json 2> gpurun_out/ab8/${w}_dev_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 8 > gpurun_out/ab8/${w}_s8_$i.json 2> gpurun_out/ab8/${w}_s8_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 1 > gpurun_out/ab8/${w}_s1_$i.json 2> gpurun_out/ab8/${w}_s1_$i.err
  done
done
python tools/ab_summary.py gpurun_out/ab8 > gpurun_out/ab8/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab8/summary.json"))
for k,v in sorted(d.items()):
    print(k, v["value"], {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in v["kernels"].items()})
PY
