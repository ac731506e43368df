# round 3: GPU suite + default bench (+ optional extra bench args as $1..): quick check of a tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c3
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/c3/gputests.log 2>&1; tail -5 gpurun_out/c3/gputests.log
timeout 400 python bench.py > gpurun_out/c3/bench.json 2> gpurun_out/c3/bench.err; tail -3 gpurun_out/c3/bench.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/c3/bench.json").read().strip().splitlines()[-1])
print("value", l["value"], "bit_exact", l["bit_exact"], l["frames_checked"], "bound", l.get("scaling_bound"), "numa", l.get("numa_node"), "mem", l.get("host_memory"), "cpu_ms", l["host_cpu"]["cpu_ms_per_frame"], "roof", l["roofline"]["kernel"], l["roofline"]["frac"], l["roofline"]["traffic_file"])
print({k:(v["avg_us"],v["pictures_per_launch"]) for k,v in l["kernels"].items()}, l.get("single_stream",{}).get("value"), l.get("device_resident_output",{}).get("value"))
PY
