# round 4: third kernel set -- one barrier per deblocking step (H of macroblock x - 1, then V of macroblock x), coherent reference loads as sc1 buffer loads
# (k_chain at 119 registers / one LDS block: four workgroups per CU) -- against the previous commit's library (scratch/_ab/v1) on one box.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab3
LAG2=$GRAFT_REPO_ROOT/jmcodec_amd/lib_dbg_lag2/libjm_amd_dec.so
timeout 1500 python -m pytest tests -m gpu -x -q -rs > gpurun_out/ab3/gputests.log 2>&1; rc=$?; tail -12 gpurun_out/ab3/gputests.log
if [ $rc -ne 0 ]; then
  JM_AMD_DEC_LIB=$LAG2 timeout 1500 python -m pytest tests -m gpu -x -q -rs > gpurun_out/ab3/gputests_lag2.log 2>&1; tail -12 gpurun_out/ab3/gputests_lag2.log
fi
python bench.py --no-extra --no-cpu-baseline --no-single --steps 3 > /dev/null 2>&1
for i in 1 2; do
  for w in v1 head lag2; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so
    [ $w = v1 ] && L=$GRAFT_REPO_ROOT/scratch/_ab/v1/jmcodec_amd/lib/libjm_amd_dec.so
    [ $w = lag2 ] && L=$LAG2
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 > gpurun_out/ab3/${w}_host_$i.json 2> gpurun_out/ab3/${w}_host_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --device-output > gpurun_out/ab3/${w}_dev_$i.json 2> gpurun_out/ab3/${w}_dev_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 8 > gpurun_out/ab3/${w}_s8_$i.json 2> gpurun_out/ab3/${w}_s8_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 16 > gpurun_out/ab3/${w}_s16_$i.json 2> gpurun_out/ab3/${w}_s16_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 1 > gpurun_out/ab3/${w}_s1_$i.json 2> gpurun_out/ab3/${w}_s1_$i.err
  done
done
python tools/ab_summary.py gpurun_out/ab3 > gpurun_out/ab3/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab3/summary.json"))
for k,v in sorted(d.items()):
    print(k, v["value"], {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in v["kernels"].items()})
PY
grep -h "gave up\|timed out\|recover" gpurun_out/ab3/*.err | sort | uniq -c | head
