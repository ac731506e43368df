# round 5: checkpoint of a tree -- the whole GPU suite, smoke(), the driver's default bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; P=gpurun_out/ck5; mkdir -p $P
timeout 2400 python -m pytest tests -m gpu -x -q -rs > $P/gputests.log 2>&1; tail -5 $P/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $P/smoke.log 2>&1; tail -3 $P/smoke.log
timeout 600 python bench.py > $P/bench.json 2> $P/bench.err; tail -2 $P/bench.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/ck5/bench.json").read().strip().splitlines()[-1])
print("value", l["value"], "bit_exact", l["bit_exact"], "bound", l.get("scaling_bound"), "cpu_ms", l["host_cpu"]["cpu_ms_per_frame"], "need8", l["host_cpu"].get("cpu_needed_for_8_gpus"), "roof", l["roofline"]["kernel"], l["roofline"]["frac"])
print({k:(v["avg_us"],v["pictures_per_launch"]) for k,v in l["kernels"].items()}, "single", l.get("single_stream",{}).get("value"), "devres", l.get("device_resident_output",{}).get("value"))
for k in ("c4_slice","c2_4k","c3_4k"):
    e=l.get(k,{}); print(k, e.get("value"), e.get("bit_exact"), e.get("scaling_bound"), e.get("host_cpu"), {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in e.get("kernels",{}).items()})
PY
