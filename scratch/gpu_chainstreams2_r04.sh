# round 4: the chain threshold at 20 / 24 / 32 streams: JM_AMD_DEC_CHAIN_STREAMS = 16 (default) / 24 / 32, device-resident and host output
cd $GRAFT_REPO_ROOT; P=gpurun_out/cs2; mkdir -p $P
for i in 1 2; do for s in 20 24 32; do for cs in 16 24 32; do for mode in dev host; do
  X=""; [ $mode = dev ] && X="--device-output"
  JM_AMD_DEC_CHAIN_STREAMS=$cs timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams $s $X > $P/cs${cs}_${mode}_s${s}_$i.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$P/cs${cs}_${mode}_s${s}_$i.json")); k=d["kernels"]
print("streams $s chain_streams $cs $mode:", d["value"], "chain launches", d["engine"]["chain_batches"], "recov", d["engine"]["chain_recoveries"], {n:(v["avg_us"],v["pictures_per_launch"]) for n,v in k.items() if n in ("k_inter","k_deblock","k_chain")})
PY
done; done; done; done
