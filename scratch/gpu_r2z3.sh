# the "direct" output route: parity under it (whole GPU suite with the route forced), then the bench against the default
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2z3
JM_AMD_DEC_OUT_FETCH=direct timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2z3/pytest_direct.txt 2>&1; tail -3 gpurun_out/r2z3/pytest_direct.txt
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2z3/$name.json 2>gpurun_out/r2z3/$name.err || echo "FAIL $name"; }
EXTRA="" run a_default_1 X=1
EXTRA="" run b_direct_1 JM_AMD_DEC_OUT_FETCH=direct
EXTRA="" run c_default_2 X=1
EXTRA="" run d_direct_2 JM_AMD_DEC_OUT_FETCH=direct
EXTRA="--streams 8" run e_s8_default X=1
EXTRA="--streams 8" run f_s8_direct JM_AMD_DEC_OUT_FETCH=direct
EXTRA="--streams 1" run g_s1_default X=1
EXTRA="--streams 1" run h_s1_direct JM_AMD_DEC_OUT_FETCH=direct
EXTRA="--streams 48" run i_s48_direct JM_AMD_DEC_OUT_FETCH=direct
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2z3/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h.get("cpu_ms_per_frame"), h["calling_threads"]["cpu_ms_per_frame"])
PY
