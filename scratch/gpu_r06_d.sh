#!/bin/bash
# round 6: the whole GPU suite on the tree with 40 job slots / chains of 16 for one or two streams / the new chain wait timers / the cheaper look for intra
# pictures; then lanes 00 vs 11 again and the stream-count ladder with the old and the new defaults
mkdir -p gpurun_out/d; O=gpurun_out/d
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/gputests.log
for cfg in "0 0" "1 1" "0 0" "1 1"; do set -- $cfg; i=$((i+1))
  JM_AMD_DEC_CROSS_LANE=$1 JM_AMD_DEC_EARLY_INTRA=$2 python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/lanes_dev_$1$2_$i.json 2> $O/lanes_dev_$1$2_$i.err
done
for s in 1 2 4 8 16; do
  python bench.py --streams $s --steps 20 --no-extra --no-cpu-baseline --no-single > $O/new_s$s.json 2> $O/new_s$s.err
  JM_AMD_DEC_JOB_SLOTS=24 JM_AMD_DEC_CHAIN_DEPTH=8 python bench.py --streams $s --steps 20 --no-extra --no-cpu-baseline --no-single > $O/old_s$s.json 2> $O/old_s$s.err
done
python bench.py > $O/bench_full.json 2> $O/bench_full.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/d/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    e = d["engine"]; ln = e.get("lanes", {})
    o = ln.get("ordinary", {}); it = ln.get("intra", {})
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], "cpus", d["host_cpu"]["cpus_busy"], "ord", o.get("pictures_per_batch"), o.get("busy_frac"), "intra", it.get("batches"), it.get("pictures_per_batch"),
          "left-out ms", ln.get("left_out_ms_per_occasion"), "early", ln.get("intra_pictures_launched_ahead_of_their_turn"), "chain", e["chain_batches_whole_run"], "pics/chain", round(e["chain_pictures_whole_run"]/max(e["chain_batches_whole_run"],1),2), "rec", e["chain_recoveries_whole_run"], "gaps", e.get("chain_launches_with_clock_gaps"), e.get("longest_clock_gap_us_whole_process"))
    if 'bench_full' in f:
        print("   single", d.get("single_stream",{}).get("value"), "c0", d.get("c0_pushpull",{}).get("value"), d.get("c0_pushpull",{}).get("vs_single_stream"), "devres", d.get("device_resident_output",{}).get("value"), d.get("device_resident_output",{}).get("scaling_bound"))
        for k in ("c4_slice","c2_4k","c3_4k"): print("   ", k, d[k].get("value"), d[k].get("scaling_bound"), d[k].get("bit_exact"), d[k].get("error"))
PY
cat $O/gputests.log
