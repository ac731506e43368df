cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zk
run() { name=$1; shift; timeout 120 env "$@" python bench.py --no-cpu-baseline --no-single --steps 3 --warmup 1 $EXTRA > gpurun_out/r2zk/$name.json 2>gpurun_out/r2zk/$name.err || echo "FAIL $name"; grep -h "gave up\|ran out" gpurun_out/r2zk/$name.err | head -3; }
EXTRA="--streams 4" run s4_split JM_AMD_DEC_VERBOSE=1
EXTRA="--streams 4" run s4_split_nohoist JM_AMD_DEC_VERBOSE=1 JM_AMD_DEC_NO_HOIST=1
EXTRA="--streams 4" run s4_split_nointra JM_AMD_DEC_VERBOSE=1 JM_AMD_DEC_NO_CHAIN_INTRA=1
EXTRA="--streams 2" run s2_split_d2 JM_AMD_DEC_VERBOSE=1 JM_AMD_DEC_CHAIN_DEPTH=2
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zk/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["engine"]["pictures_per_batch"], d["kernels"]["k_chain"])
PY
