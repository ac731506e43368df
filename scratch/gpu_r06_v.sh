#!/bin/bash
# round 6: how many hardware queues the HIP runtime multiplexes the engine's streams onto (GPU_MAX_HW_QUEUES, default 4): 4 / 8 / 16, per leg, alternating
mkdir -p gpurun_out/v; O=gpurun_out/v
for i in 1 2; do
  for q in 4 8 16; do
    GPU_MAX_HW_QUEUES=$q python bench.py --codec hevc --width 3840 --height 2160 --streams 16 --frames 16 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c3_q${q}_$i.json 2> $O/c3_q${q}_$i.err
    GPU_MAX_HW_QUEUES=$q python bench.py --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c2_q${q}_$i.json 2> $O/c2_q${q}_$i.err
    GPU_MAX_HW_QUEUES=$q python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/dev_q${q}_$i.json 2> $O/dev_q${q}_$i.err
    GPU_MAX_HW_QUEUES=$q python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_q${q}_$i.json 2> $O/host_q${q}_$i.err
    GPU_MAX_HW_QUEUES=$q python bench.py --streams 1 --steps 20 --no-extra --no-cpu-baseline --no-single > $O/s1_q${q}_$i.json 2> $O/s1_q${q}_$i.err
    GPU_MAX_HW_QUEUES=$q python bench.py --streams 8 --no-extra --no-cpu-baseline --no-single > $O/s8_q${q}_$i.json 2> $O/s8_q${q}_$i.err
  done
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/v/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    ln = d["engine"].get("lanes", {}); o = ln.get("ordinary", ln.get("hevc", {}))
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], d.get("bound_utilisation"), "cpu ms/frame", d["host_cpu"]["cpu_ms_per_frame"], "lane", o.get("pictures_per_batch"), o.get("busy_frac"), "job lists", o.get("idle_waiting_for_job_lists_frac"), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
PY
