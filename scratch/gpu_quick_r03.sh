# round 3: H.264 GPU parity tests only + a short default bench and a one-stream bench (kernel timings)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q3
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/q3/gputests.log 2>&1; tail -4 gpurun_out/q3/gputests.log
for a in "" "--streams 1" "--streams 8"; do
timeout 400 python bench.py $a --no-cpu-baseline --steps 6 > gpurun_out/q3/bench.json 2> gpurun_out/q3/bench.err; tail -2 gpurun_out/q3/bench.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/q3/bench.json").read().strip().splitlines()[-1])
print("value", l["value"], "bit_exact", l["bit_exact"], "cpu_ms", l["host_cpu"]["cpu_ms_per_frame"], "roof", l["roofline"]["kernel"], l["roofline"]["frac"], "mem", l["host_memory"]["job_slots_mb"], l["host_memory"]["peak_rss_mb"])
print({k:(v["avg_us"],v["pictures_per_launch"]) for k,v in l["kernels"].items()}, l.get("single_stream",{}).get("value"), l.get("device_resident_output",{}).get("value"))
PY
done
