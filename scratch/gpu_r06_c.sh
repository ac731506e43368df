#!/bin/bash
# round 6, one call: (a) lanes with intra pictures ahead of their turn once their upload has landed; (b) one stream: job slots x chain depth;
# (c) which output route makes a chain launch's waves stand still (clock gaps seen by the waits)
mkdir -p gpurun_out/c; O=gpurun_out/c
for cfg in "0 0" "1 1" "0 0" "1 1"; do set -- $cfg; i=$((i+1))
  JM_AMD_DEC_CROSS_LANE=$1 JM_AMD_DEC_EARLY_INTRA=$2 python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/lanes_dev_$1$2_$i.json 2> $O/lanes_dev_$1$2_$i.err
done
JM_AMD_DEC_LANE_TRACE=20000 python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 4 > $O/trace_11.json 2> $O/trace_11.err
for slots in 24 40 56; do for depth in 8 12 16; do
  JM_AMD_DEC_JOB_SLOTS=$slots JM_AMD_DEC_CHAIN_DEPTH=$depth python bench.py --streams 1 --steps 20 --no-extra --no-cpu-baseline --no-single > $O/s1_j${slots}_d${depth}.json 2> $O/s1_j${slots}_d${depth}.err
done; done
for r in 1 2 3 4; do
  JM_AMD_DEC_VERBOSE=1 python bench.py --streams 2 --steps 20 --no-extra --no-cpu-baseline --no-single > $O/gap_direct_$r.json 2> $O/gap_direct_$r.err
  JM_AMD_DEC_VERBOSE=1 JM_AMD_DEC_OUT_PINNED=1 python bench.py --streams 2 --steps 20 --no-extra --no-cpu-baseline --no-single > $O/gap_pinned_$r.json 2> $O/gap_pinned_$r.err
  JM_AMD_DEC_VERBOSE=1 python bench.py --device-output --streams 2 --steps 20 --no-extra --no-cpu-baseline --no-single > $O/gap_devout_$r.json 2> $O/gap_devout_$r.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/c/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    e = d["engine"]; ln = e.get("lanes", {})
    o = ln.get("ordinary", {}); it = ln.get("intra", {})
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], "cpus", d["host_cpu"]["cpus_busy"], "ord", o.get("pictures_per_batch"), o.get("busy_frac"), "intra", it.get("batches"), it.get("pictures_per_batch"),
          "left-out ms", ln.get("left_out_ms_per_occasion"), "early", ln.get("intra_pictures_launched_ahead_of_their_turn"), "chain", e["chain_batches_whole_run"], "pics/chain", round(e["chain_pictures_whole_run"]/max(e["chain_batches_whole_run"],1),2), "rec", e["chain_recoveries_whole_run"], "gaps", e.get("chain_launches_with_clock_gaps"), e.get("longest_clock_gap_us_whole_process"))
PY
grep -c "clock gap" $O/gap_*.err
grep -h "clock gap" $O/gap_*.err | sort | uniq -c | head -20
