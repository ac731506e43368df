#!/bin/bash
# round 6: the engine loop's walk (ends when every decoder with pending pictures was seen) and the engine thread's timer slack, beside the library before both (d62a2d5):
#   old  = scratch/_ab/d62 library;  s50 = new library, JM_AMD_DEC_TIMER_SLACK_NS=50000 (the default slack: 20 us sleeps take ~75);  s1 = new library, 1 us slack (default)
mkdir -p gpurun_out/m; O=gpurun_out/m
OLD=$PWD/scratch/_ab/d62/jmcodec_amd/lib/libjm_amd_dec.so
for i in 1 2 3; do
  for cfg in old s50 s1; do
    case $cfg in old) E="JM_AMD_DEC_LIB=$OLD";; s50) E="JM_AMD_DEC_TIMER_SLACK_NS=50000";; s1) E="JM_AMD_DEC_TIMER_SLACK_NS=1000";; esac
    env $E python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_${cfg}_$i.json 2> $O/dev_${cfg}_$i.err
    env $E python bench.py --streams 1 --steps 20 --no-extra --no-cpu-baseline --no-single > $O/s1_${cfg}_$i.json 2> $O/s1_${cfg}_$i.err
    env $E python bench.py --streams 8 --no-extra --no-cpu-baseline --no-single > $O/s8_${cfg}_$i.json 2> $O/s8_${cfg}_$i.err
    env $E python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_${cfg}_$i.json 2> $O/host_${cfg}_$i.err
  done
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/m/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    e = d["engine"]; ln = e.get("lanes", {}); o = ln.get("ordinary", {}); r = d["roofline"]; bt = d["host_cpu"].get("by_thread", {})
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], "cpus", d["host_cpu"]["cpus_busy"], "engine thread s", bt.get("jm-engine", {}).get("user_s"), "ord", o.get("pictures_per_batch"), o.get("busy_frac"), o.get("idle_between_batches_frac"), "roofline", r["kernel"], r["frac"], r["pictures_per_launch"])
PY
