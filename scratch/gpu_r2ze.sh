cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2ze
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2ze/$name.json 2>gpurun_out/r2ze/$name.err || echo "FAIL $name"; }
for e in 8 4 4,8 2,4,8 2,8 1,4,8 1,8 2,4,8; do
EXTRA="--steps 20" run "b_direct_$(echo $e | tr -d ,)_$RANDOM" JM_AMD_DEC_OUT_FETCH=direct JM_AMD_DEC_COPY_ENGINES=$e
done
EXTRA="--steps 20 --streams 48" run c_s48_248 JM_AMD_DEC_OUT_FETCH=direct JM_AMD_DEC_COPY_ENGINES=2,4,8
EXTRA="--steps 20 --streams 16" run c_s16_248 JM_AMD_DEC_OUT_FETCH=direct JM_AMD_DEC_COPY_ENGINES=2,4,8
EXTRA="--steps 20 --streams 8" run c_s8_248 JM_AMD_DEC_OUT_FETCH=direct JM_AMD_DEC_COPY_ENGINES=2,4,8
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2ze/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], h["cpus_busy"], h.get("cpu_ms_per_frame"), h["calling_threads"]["cpu_ms_per_frame"], d["engine"]["pictures_per_batch"], d["engine"]["direct_output"]["caller_wait_us_per_frame"])
PY
