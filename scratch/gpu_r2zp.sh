cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2zp/pytest.txt 2>&1; tail -3 gpurun_out/r2zp/pytest.txt
run() { name=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2zp/$name.json 2>gpurun_out/r2zp/$name.err || echo "FAIL $name"; }
EXTRA="--tools high_b" run high_b X=1
EXTRA="--tools high_b --streams 4" run high_b_s4 X=1
EXTRA="--tools high" run high X=1
EXTRA="" run base X=1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zp/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], d["kernels"]["k_inter"], d["kernels"]["k_chain"])
PY
