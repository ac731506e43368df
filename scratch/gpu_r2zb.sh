cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zb
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2zb/$name.json 2>gpurun_out/r2zb/$name.err || echo "FAIL $name"; }
EXTRA="--steps 20" run a_default_k20 X=1
EXTRA="--steps 20" run b_direct_k20 JM_AMD_DEC_OUT_FETCH=direct
EXTRA="--steps 20" run b_direct_d4_k20 JM_AMD_DEC_OUT_FETCH=direct JM_AMD_DEC_DISPLAY_DELAY=4
EXTRA="--steps 20 --device-output" run c_dev_k20 X=1
EXTRA="--steps 20 --device-output" run c_dev_d4_k20 JM_AMD_DEC_DISPLAY_DELAY=4
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zb/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], h["cpus_busy"], h.get("cpu_ms_per_frame"), d["engine"]["pictures_per_batch"], d["engine"]["formation"])
PY
