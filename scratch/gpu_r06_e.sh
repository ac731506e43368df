#!/bin/bash
# round 6: HEVC kernels on packed arithmetic (parity first), the intra lane one batch at a time, the push / pull hold-off by pipeline depth
mkdir -p gpurun_out/e; O=gpurun_out/e
python -m pytest tests/test_hevc_gpu_parity.py tests/test_pushpull.py -m gpu -x -q 2>&1 | tail -12 > $O/hevc_tests.log
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "intel or concurrent or c4_slice or full_size or b_streams or all_intra" 2>&1 | tail -8 > $O/h264_subset.log
for cfg in "0 0" "1 1" "0 0" "1 1" "1 1"; do set -- $cfg; i=$((i+1))
  JM_AMD_DEC_CROSS_LANE=$1 JM_AMD_DEC_EARLY_INTRA=$2 python bench.py --device-output --no-extra --no-cpu-baseline --no-single --steps 10 > $O/lanes_dev_$1$2_$i.json 2> $O/lanes_dev_$1$2_$i.err
done
JM_AMD_DEC_CROSS_LANE=0 JM_AMD_DEC_EARLY_INTRA=0 python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_00.json 2> $O/host_00.err
python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_11.json 2> $O/host_11.err
python bench.py > $O/bench_full.json 2> $O/bench_full.err
python bench.py --codec hevc --width 3840 --height 2160 --streams 16 --frames 16 --steps 3 --no-extra --no-cpu-baseline --no-single --device-output > $O/c3_dev.json 2> $O/c3_dev.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/e/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    e = d["engine"]; ln = e.get("lanes", {})
    o = ln.get("ordinary", {}); it = ln.get("intra", {})
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], "cpus", d["host_cpu"]["cpus_busy"], "ord", o.get("pictures_per_batch"), o.get("busy_frac"), "intra", it.get("batches"), it.get("pictures_per_batch"),
          "left-out ms", ln.get("left_out_ms_per_occasion"), "early", ln.get("intra_pictures_launched_ahead_of_their_turn"), "rec", e["chain_recoveries_whole_run"], "gaps", e.get("chain_launches_with_clock_gaps"), e.get("longest_clock_gap_us_whole_process"))
    print("    kernels", {k: (v["launches"], v["avg_us"], v["pictures_per_launch"]) for k, v in d["kernels"].items() if v["launches"]})
    if 'bench_full' in f:
        print("   single", d.get("single_stream",{}).get("value"), "c0", d.get("c0_pushpull",{}).get("value"), d.get("c0_pushpull",{}).get("vs_single_stream"), "devres", d.get("device_resident_output",{}).get("value"), d.get("device_resident_output",{}).get("scaling_bound"))
        for k in ("c4_slice","c2_4k","c3_4k"): print("   ", k, d[k].get("value"), d[k].get("scaling_bound"), d[k].get("bit_exact"), d[k].get("error"), d[k].get("kernels"))
PY
cat $O/hevc_tests.log $O/h264_subset.log
