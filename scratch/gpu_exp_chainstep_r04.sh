# round 4, measurement builds (results are NOT checked -- the variants drop ordering the product needs): what do the counted wait in front of `fin` and the
# write-through stores cost a deblocking step of a chain launch?  1 / 8 streams, k_chain per launch and frames/s; the time line of a one-stream launch.
cd $GRAFT_REPO_ROOT; P=gpurun_out/exp1; mkdir -p $P
for i in 1 2; do for w in head nofin nowt nowtfin; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w != head ] && L=$GRAFT_REPO_ROOT/jmcodec_amd/lib_dbg_$w/libjm_amd_dec.so
  for s in 1 8; do
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams $s > $P/${w}_s${s}_$i.json 2>/dev/null
    python - <<PY
import json
try:
    d=json.load(open("$P/${w}_s${s}_$i.json")); k=d["kernels"].get("k_chain",{})
    print("$w streams $s:", d["value"], "bit_exact", d.get("bit_exact"), "k_chain", k.get("avg_us"), k.get("pictures_per_launch"), "recov", d["engine"]["chain_recoveries"])
except Exception as e: print("$w streams $s: failed", e)
PY
  done
done; done
for w in head nowtfin; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w != head ] && L=$GRAFT_REPO_ROOT/jmcodec_amd/lib_dbg_$w/libjm_amd_dec.so
  JM_AMD_DEC_LIB=$L JM_AMD_DEC_CENSUS=1 JM_AMD_DEC_CHAIN_TIMELINE=1 timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 2 --warmup 1 --streams 1 > /dev/null 2> $P/tl_$w.err
  echo "== $w"; awk '/chain launch of 8/{n++} n==3' $P/tl_$w.err | grep -E "time line|reconstruction workgroups" | cut -c1-160
done
