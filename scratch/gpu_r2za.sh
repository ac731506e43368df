cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2za
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2za/$name.json 2>gpurun_out/r2za/$name.err || echo "FAIL $name"; }
EXTRA="" run a_default_d0 X=1
EXTRA="" run b_direct_d0 JM_AMD_DEC_OUT_FETCH=direct
EXTRA="" run b_direct_d3 JM_AMD_DEC_OUT_FETCH=direct JM_AMD_DEC_DISPLAY_DELAY=3
EXTRA="" run b_direct_d3_nochain JM_AMD_DEC_OUT_FETCH=direct JM_AMD_DEC_DISPLAY_DELAY=3 JM_AMD_DEC_CHAIN_STREAMS=0
EXTRA="--device-output" run c_dev X=1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2za/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], h["cpus_busy"], h.get("cpu_ms_per_frame"), d["engine"]["pictures_per_batch"], d["engine"]["formation"], d.get("host_diag"))
PY
