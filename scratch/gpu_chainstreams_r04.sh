# round 4: up to how many active streams should chain launches be formed?  The default workload (32 streams; the active count dips below the threshold while
# callers wait for their frames) with JM_AMD_DEC_CHAIN_STREAMS = 16 (default) / 12 / 8 / 4 / 0, host output and device-resident, and 16 / 12 streams.
cd $GRAFT_REPO_ROOT; P=gpurun_out/cs; mkdir -p $P
for i in 1 2; do for cs in 16 12 8 4 0; do for mode in host dev; do for s in 32 16 12; do
  X=""; [ $mode = dev ] && X="--device-output"
  [ $mode = host ] && [ $s != 32 ] && continue
  JM_AMD_DEC_CHAIN_STREAMS=$cs timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams $s $X > $P/cs${cs}_${mode}_s${s}_$i.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$P/cs${cs}_${mode}_s${s}_$i.json")); k=d["kernels"]
print("chain_streams $cs $mode streams $s:", d["value"], "chain launches", d["engine"]["chain_batches"], {n:(v["avg_us"],v["pictures_per_launch"]) for n,v in k.items() if n in ("k_inter","k_deblock","k_chain")})
PY
done; done; done; done
