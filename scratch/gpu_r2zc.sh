cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zc
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2zc/$name.json 2>gpurun_out/r2zc/$name.err || echo "FAIL $name"; }
EXTRA="--steps 20" run b_direct_k20 JM_AMD_DEC_OUT_FETCH=direct
EXTRA="--steps 20 --streams 8" run b_direct_s8 JM_AMD_DEC_OUT_FETCH=direct
EXTRA="--steps 20 --streams 1" run b_direct_s1 JM_AMD_DEC_OUT_FETCH=direct
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zc/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], h["cpus_busy"], h.get("cpu_ms_per_frame"), h["calling_threads"]["cpu_ms_per_frame"], d["engine"]["pictures_per_batch"], d["engine"]["formation"], d["engine"]["direct_output"])
PY
