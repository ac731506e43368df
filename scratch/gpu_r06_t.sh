#!/bin/bash
# round 6: is C3 (HEVC 4K, 16 streams) slower with two copy streams, or was the final call's box slow?  Alternating, three runs each; plus C2 once each.
mkdir -p gpurun_out/t; O=gpurun_out/t
for i in 1 2 3; do
  for c in 1 2; do
    JM_AMD_DEC_COPY_STREAMS=$c python bench.py --codec hevc --width 3840 --height 2160 --streams 16 --frames 16 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/c3_c${c}_$i.json 2> $O/c3_c${c}_$i.err
  done
done
for c in 1 2; do
  JM_AMD_DEC_COPY_STREAMS=$c python bench.py --codec hevc --width 1920 --height 1080 --streams 16 --frames 32 --steps 3 --no-extra --no-cpu-baseline --no-single > $O/h1080_c${c}_1.json 2> $O/h1080_c${c}_1.err
done
python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_final_1.json 2> $O/host_final_1.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/t/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    ln = d["engine"].get("lanes", {}); o = ln.get("hevc", ln.get("ordinary", {}))
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], d.get("bound_utilisation"), "cpu ms/frame", d["host_cpu"]["cpu_ms_per_frame"], "lane", o.get("pictures_per_batch"), o.get("busy_frac"), o.get("idle_waiting_for_job_lists_frac"), o.get("batches_launched_after_the_lane_ran_dry"), o.get("batches"))
PY
