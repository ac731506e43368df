# round 6, final tree (two high-priority copy streams, fill linger): a short soak of the chain launches -- 1 / 2 / 3 / 4 / 6 / 8 / 12 / 16 streams, once with the defaults and
# once with varied chain knobs, C2 with chains forced on x 2, High + B at 4 streams: every line must be bit_exact with 0 recoveries and 0 device wait errors
cd $GRAFT_REPO_ROOT; P=gpurun_out/soak7; mkdir -p $P
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs JM_AMD_DEC_VERBOSE=1 timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single "$@" > $P/$tag.json 2> $P/$tag.err; }
for r in 1 3; do for s in 1 2 3 4 6 8 12 16; do
  E=""; [ $r = 3 ] && E="JM_AMD_DEC_CHAIN_DEPTH=$((2 + (s * 7) % 7)) JM_AMD_DEC_CHAIN_LAG=$((24 + (s * 13) % 40))"
  run c1_s${s}_r$r $E -- --steps 20 --streams $s
done; done
for i in 1 2; do run c2_chain_$i JM_AMD_DEC_CHAIN_STREAMS=64 -- --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3; done
run highb_s4_1 -- --tools high_b --streams 4 --steps 10; run high_s2_1 -- --tools high --streams 2 --steps 10
python - <<'PY'
import json, glob, os
tot_b = tot_p = tot_r = bad = 0
for f in sorted(glob.glob('gpurun_out/soak7/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(os.path.basename(f), 'NO LINE'); bad += 1; continue
    e = d["engine"]
    tot_b += e["chain_batches_whole_run"]; tot_p += e["chain_pictures_whole_run"]; tot_r += e["chain_recoveries_whole_run"]
    ok = d["bit_exact"] and e["chain_recoveries_whole_run"] == 0 and e["device_wait_errors"] == 0 and d["decode_errors"] == 0
    bad += not ok
    print(os.path.basename(f), d["value"], "bit_exact", d["bit_exact"], "chain launches", e["chain_batches_whole_run"], "pictures per launch", round(e["chain_pictures_whole_run"] / max(e["chain_batches_whole_run"], 1), 1),
          "recoveries", e["chain_recoveries_whole_run"], "wait errors", e["device_wait_errors"], "launches with clock gaps", e.get("chain_launches_with_clock_gaps"), "longest gap us", e.get("longest_clock_gap_us_whole_process"))
print("TOTAL chain launches", tot_b, "pictures in them", tot_p, "recoveries", tot_r, "lines not ok", bad)
PY
grep -h "clock gap\|gave up" $P/*.err | sort | uniq -c | sort -rn | head -8
