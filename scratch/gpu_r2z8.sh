# display delay x output route
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2z9
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2z9/$name.json 2>gpurun_out/r2z9/$name.err || echo "FAIL $name"; }
for d in 0 1 2 3 5; do
EXTRA="" run a_default_d$d JM_AMD_DEC_DISPLAY_DELAY=$d
EXTRA="" run b_direct_d$d JM_AMD_DEC_OUT_FETCH=direct JM_AMD_DEC_DISPLAY_DELAY=$d
done
EXTRA="--streams 8" run f_s8_direct_d4 JM_AMD_DEC_OUT_FETCH=direct JM_AMD_DEC_DISPLAY_DELAY=4
EXTRA="--streams 1" run h_s1_direct_d4 JM_AMD_DEC_OUT_FETCH=direct JM_AMD_DEC_DISPLAY_DELAY=4
EXTRA="--streams 1" run h_s1_default_d4 JM_AMD_DEC_DISPLAY_DELAY=4
EXTRA="--streams 16" run j_s16_direct_d4 JM_AMD_DEC_OUT_FETCH=direct JM_AMD_DEC_DISPLAY_DELAY=4
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2z9/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h.get("cpu_ms_per_frame"), h["calling_threads"]["cpu_ms_per_frame"], d["engine"]["pictures_per_batch"])
PY
