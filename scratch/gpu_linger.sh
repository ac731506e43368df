# single / dual stream: chain depth (JM_AMD_DEC_CHAIN_DEPTH) with the "wait for the running chain launch" rule on
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/lg
for v in 8 12 8 12; do for s in 1 2 8; do
JM_AMD_DEC_CHAIN_DEPTH=$v timeout 300 python bench.py --streams $s --no-cpu-baseline --no-single > gpurun_out/lg/b.json 2>/dev/null
python - $v $s <<'PY'
import json, sys
l=json.loads(open("gpurun_out/lg/b.json").read().strip().splitlines()[-1])
k=l["kernels"]["k_chain"]
print("depth", sys.argv[1], "streams", sys.argv[2], "value", l["value"], l["bit_exact"], "k_chain", k["avg_us"], k["pictures_per_launch"], k["launches"], l["engine"]["formation"])
PY
done; done
