#!/bin/bash
# round 5: the soak of the quad-window tree had 2 chain launches give up (c1, 16 and 6 streams; ring waits of bands of pictures that were reconstructed before the
# launch).  Which change?  The same two shapes, interleaved over three libraries: head (XCD-ordered work list + quad windows), prev2 (XCD order only), prev (neither).
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/bis; mkdir -p $P
for i in $(seq 1 ${ROUNDS:-10}); do for w in ${LIBS:-head prev2 prev}; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w != head ] && L=$GRAFT_REPO_ROOT/scratch/_ab/$w/libjm_amd_dec.so
  for s in 16 6; do
    E=""; [ $((i % 2)) = 0 ] && E="JM_AMD_DEC_CHAIN_DEPTH=$((2 + (s * 7 + i) % 7)) JM_AMD_DEC_CHAIN_LAG=$((24 + (s * 13 + i) % 40))"
    env $E JM_AMD_DEC_LIB=$L JM_AMD_DEC_VERBOSE=1 timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --streams $s > $P/${w}_s${s}_$i.json 2> $P/${w}_s${s}_$i.err
  done
done; done
python3 - <<'PY'
import json, glob, os, collections
tot = collections.defaultdict(lambda: [0, 0, 0, 0])
for f in sorted(glob.glob('gpurun_out/bis/*.json')):
    w = os.path.basename(f).split('_')[0]
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: tot[w][3] += 1; continue
    e = d["engine"]
    tot[w][0] += 1; tot[w][1] += e["chain_batches_whole_run"]; tot[w][2] += e["chain_recoveries_whole_run"]
    if e["chain_recoveries_whole_run"] or not d["bit_exact"]: print(os.path.basename(f), "recoveries", e["chain_recoveries_whole_run"], "bit_exact", d["bit_exact"])
for w, (n, b, r, bad) in tot.items(): print(w, "runs", n, "chain launches", b, "recoveries", r, "unreadable", bad)
PY
grep -h "FIRST give-up" $P/*.err | cut -c1-40,290-420 | sort | uniq -c | head
echo finished
