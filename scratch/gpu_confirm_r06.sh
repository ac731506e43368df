#!/bin/bash
# round 6, last call: what the driver runs, on the final tree -- GPU suite, smoke(), the default bench line; plus two short headline runs for the engine thread's CPU time
mkdir -p gpurun_out/confirm6; O=gpurun_out/confirm6
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/gputests.log; cat $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > $O/r06_bench_confirm.json 2> $O/r06_bench_confirm.err
for i in 1 2; do python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_$i.json 2> $O/host_$i.err; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/confirm6/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    o = d["engine"]["lanes"].get("ordinary", {}); it = d["engine"]["lanes"].get("intra", {})
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], d.get("bound_utilisation"), "engine thread", d["host_cpu"]["by_thread"].get("jm-engine"), "ord", o.get("pictures_per_batch"), o.get("busy_frac"), "intra", it.get("pictures_per_batch"), "roofline", d["roofline"]["frac"])
    for k in ('single_stream','c0_pushpull','device_resident_output','c4_slice','c2_4k','c3_4k'):
        v = d.get(k)
        if v: print("   ", k, v.get("value"), v.get("scaling_bound"), v.get("bit_exact"), v.get("vs_single_stream"))
PY
