# round 4: the new kernels (deblocking wavefront with one step per row, Intra4x4 edge path in registers, frame-only reconstruction instantiation) on one box:
# GPU suite (the default library; on failure also the variant with the old two-steps-per-row deblocking, to tell the changes apart), then bench lines of
# four libraries side by side -- round 2 (edb9cea), round 3 (6d7fa76), HEAD, HEAD with JM_DEBLOCK_ROW_LAG=2 -- same bench.py, same streams.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab2
LAG2=$GRAFT_REPO_ROOT/jmcodec_amd/lib_dbg_lag2/libjm_amd_dec.so
timeout 1500 python -m pytest tests -m gpu -x -q -rs > gpurun_out/ab2/gputests.log 2>&1; rc=$?; tail -12 gpurun_out/ab2/gputests.log
if [ $rc -ne 0 ]; then
  JM_AMD_DEC_LIB=$LAG2 timeout 1500 python -m pytest tests -m gpu -x -q -rs > gpurun_out/ab2/gputests_lag2.log 2>&1; tail -12 gpurun_out/ab2/gputests_lag2.log
fi
python bench.py --no-extra --no-cpu-baseline --no-single --steps 3 > /dev/null 2>&1
for i in 1 2; do
  for w in r2 r3 head lag2; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so
    [ $w = r2 ] && L=$GRAFT_REPO_ROOT/scratch/_ab/r2/jmcodec_amd/lib/libjm_amd_dec.so
    [ $w = r3 ] && L=$GRAFT_REPO_ROOT/scratch/_ab/r3/jmcodec_amd/lib/libjm_amd_dec.so
    [ $w = lag2 ] && L=$LAG2
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 > gpurun_out/ab2/${w}_host_$i.json 2> gpurun_out/ab2/${w}_host_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --device-output > gpurun_out/ab2/${w}_dev_$i.json 2> gpurun_out/ab2/${w}_dev_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 8 > gpurun_out/ab2/${w}_s8_$i.json 2> gpurun_out/ab2/${w}_s8_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 1 > gpurun_out/ab2/${w}_s1_$i.json 2> gpurun_out/ab2/${w}_s1_$i.err
  done
done
python tools/ab_summary.py gpurun_out/ab2 > gpurun_out/ab2/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab2/summary.json"))
for k,v in sorted(d.items()):
    print(k, v["value"], {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in v["kernels"].items()})
PY
timeout 900 python bench.py > gpurun_out/ab2/bench_default.json 2> gpurun_out/ab2/bench_default.err; tail -3 gpurun_out/ab2/bench_default.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/ab2/bench_default.json").read().strip().splitlines()[-1])
print("value", l["value"], "bit_exact", l["bit_exact"], l["frames_checked"], "bound", l.get("scaling_bound"), "cpu_ms", l["host_cpu"]["cpu_ms_per_frame"], "roof", l["roofline"]["kernel"], l["roofline"]["frac"])
print({k:(v["avg_us"],v["pictures_per_launch"]) for k,v in l["kernels"].items()}, l.get("single_stream",{}).get("value"), l.get("device_resident_output",{}).get("value"))
for k in ("c4_slice","c2_4k","c3_4k"):
    x=l.get(k) or {}
    print(k, x.get("value"), x.get("bit_exact"), x.get("scaling_bound"), x.get("roofline"), x.get("kernels"), x.get("engine"))
PY
