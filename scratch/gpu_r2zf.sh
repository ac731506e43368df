# direct route as the default: whole GPU suite, then the bench at several stream counts
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zf
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2zf/pytest.txt 2>&1; tail -3 gpurun_out/r2zf/pytest.txt
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2zf/$name.json 2>gpurun_out/r2zf/$name.err || echo "FAIL $name"; }
EXTRA="" run a_default_1 JM_AMD_DEC_VERBOSE=1
EXTRA="" run a_default_2 X=1
EXTRA="--steps 20" run a_default_k20 X=1
EXTRA="--streams 1" run b_s1 X=1
EXTRA="--streams 4" run b_s4 X=1
EXTRA="--streams 8" run b_s8 X=1
EXTRA="--streams 16" run b_s16 X=1
EXTRA="--streams 16" run b_s16_cs16 JM_AMD_DEC_CHAIN_STREAMS=16
EXTRA="--streams 16" run b_s16_cs20 JM_AMD_DEC_CHAIN_STREAMS=20
EXTRA="--streams 12" run b_s12 X=1
EXTRA="--steps 20" run c_old_2_5 JM_AMD_DEC_OUT_FETCH=2/5
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zf/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h.get("cpu_ms_per_frame"), d["engine"]["pictures_per_batch"], d["engine"]["direct_output"])
PY
grep -h "SDMA engines" gpurun_out/r2zf/*.err | head -2
