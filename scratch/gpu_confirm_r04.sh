# round 4, last call: what the driver runs at round end, on the final tree -- GPU suite, smoke(), the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/cf
timeout 1500 python -m pytest tests -m gpu -q -rs > gpurun_out/cf/gputests.log 2>&1; tail -5 gpurun_out/cf/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
S=$(date +%s); timeout 900 python bench.py > gpurun_out/cf/bench.json 2> gpurun_out/cf/bench.err; echo "bench wall $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/cf/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["scaling_bound"], d["bit_exact"], d["roofline"]["kernel"], d["roofline"]["frac"], "single", d["single_stream"]["value"], "device-resident", d["device_resident_output"]["value"])
for k in ("c4_slice","c2_4k","c3_4k"): print(k, d[k]["value"], d[k]["bit_exact"], d[k]["scaling_bound"])
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "extra keys", [k for k in d if k.startswith("c")][:8])
PY
