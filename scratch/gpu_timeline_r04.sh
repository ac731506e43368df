# round 4: time line of chain launches (JM_AMD_DEC_CENSUS + JM_AMD_DEC_CHAIN_TIMELINE): when do the pictures of a launch start and end, how long does a
# reconstruction workgroup live and wait; then the chain depth (pictures of one stream per launch) at 1 / 2 / 4 streams
cd $GRAFT_REPO_ROOT; P=gpurun_out/tl; mkdir -p $P
for s in 1 8; do
  JM_AMD_DEC_CENSUS=1 JM_AMD_DEC_CHAIN_TIMELINE=1 timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 2 --warmup 1 --streams $s > $P/s$s.json 2> $P/s$s.err
  grep -c "chain launch of" $P/s$s.err
  grep "reconstruction workgroups:" $P/s$s.err | sort | uniq -c | sort -rn | head -5
done
for i in 1 2; do for s in 1 2 4; do for d in 8 12 16; do
  JM_AMD_DEC_CHAIN_DEPTH=$d timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams $s > $P/d${d}_s${s}_$i.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$P/d${d}_s${s}_$i.json")); k=d["kernels"].get("k_chain",{})
print("depth $d streams $s:", d["value"], "k_chain", k.get("avg_us"), k.get("pictures_per_launch"))
PY
done; done; done
