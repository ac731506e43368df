# round 4: the work-list lag between consecutive pictures of a stream (JM_AMD_DEC_CHAIN_LAG, keys) at 4 / 8 streams and device-resident 16 streams
cd $GRAFT_REPO_ROOT; P=gpurun_out/lag; mkdir -p $P
for i in 1 2; do for s in 4 8 16; do for lag in 24 48 96 160; do
  X=""; [ $s = 16 ] && X="--device-output"
  JM_AMD_DEC_CHAIN_LAG=$lag timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams $s $X > $P/l${lag}_s${s}_$i.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$P/l${lag}_s${s}_$i.json")); k=d["kernels"].get("k_chain",{})
print("lag $lag streams $s:", d["value"], "k_chain", k.get("avg_us"), k.get("pictures_per_launch"))
PY
done; done; done
