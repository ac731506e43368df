# round 4, item 2: the round-2 tree's library (scratch/_ab/r2, edb9cea) against HEAD's on ONE box, same bench.py, same streams (JM_AMD_DEC_LIB swaps only the
# library), runs alternating; then the GPU suite and the default bench line with its new extra legs.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
R2=$GRAFT_REPO_ROOT/scratch/_ab/r2/jmcodec_amd/lib/libjm_amd_dec.so
HEAD=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so
python bench.py --no-extra --no-cpu-baseline --no-single --steps 3 > /dev/null 2>&1     # generate + cache the streams, page in torch
for i in 1 2 3; do
  for w in r2 head; do
    L=$HEAD; [ $w = r2 ] && L=$R2
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 > gpurun_out/ab/${w}_host_$i.json 2> gpurun_out/ab/${w}_host_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --device-output > gpurun_out/ab/${w}_dev_$i.json 2> gpurun_out/ab/${w}_dev_$i.err
  done
done
python tools/ab_summary.py gpurun_out/ab > gpurun_out/ab/summary.json; cat gpurun_out/ab/summary.json
timeout 2400 python -m pytest tests -m gpu -x -q -rs > gpurun_out/ab/gputests.log 2>&1; tail -8 gpurun_out/ab/gputests.log
timeout 900 python bench.py > gpurun_out/ab/bench_default.json 2> gpurun_out/ab/bench_default.err; tail -3 gpurun_out/ab/bench_default.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/ab/bench_default.json").read().strip().splitlines()[-1])
print("value", l["value"], "bit_exact", l["bit_exact"], l["frames_checked"], "bound", l.get("scaling_bound"), "cpu_ms", l["host_cpu"]["cpu_ms_per_frame"], "roof", l["roofline"]["kernel"], l["roofline"]["frac"])
print({k:(v["avg_us"],v["pictures_per_launch"]) for k,v in l["kernels"].items()}, l.get("single_stream",{}).get("value"), l.get("device_resident_output",{}).get("value"))
for k in ("c4_slice","c2_4k","c3_4k"):
    print(k, json.dumps(l.get(k)))
PY
