# the "direct" output route over ROCr (host_copy.cpp): parity under it, then the bench against the default
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2z7
JM_AMD_DEC_OUT_FETCH=direct timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c4_slice or resolution or api_protocol or concurrent or chunking or harness" > gpurun_out/r2z7/pytest_direct.txt 2>&1; tail -3 gpurun_out/r2z7/pytest_direct.txt
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2z7/$name.json 2>gpurun_out/r2z7/$name.err || echo "FAIL $name"; }
EXTRA="" run a_default_1 X=1
EXTRA="" run b_direct_1 JM_AMD_DEC_OUT_FETCH=direct
EXTRA="" run c_direct_2 JM_AMD_DEC_OUT_FETCH=direct
EXTRA="--streams 8" run f_s8_direct JM_AMD_DEC_OUT_FETCH=direct
EXTRA="--streams 1" run h_s1_direct JM_AMD_DEC_OUT_FETCH=direct
EXTRA="--streams 48" run i_s48_direct JM_AMD_DEC_OUT_FETCH=direct
EXTRA="--streams 16" run j_s16_direct JM_AMD_DEC_OUT_FETCH=direct
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2z7/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h.get("cpu_ms_per_frame"), h["calling_threads"]["cpu_ms_per_frame"], d["engine"]["pictures_per_batch"])
PY

