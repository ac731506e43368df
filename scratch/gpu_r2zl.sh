cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zl
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2zl/pytest.txt 2>&1; tail -3 gpurun_out/r2zl/pytest.txt
run() { name=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2zl/$name.json 2>gpurun_out/r2zl/$name.err || echo "FAIL $name"; }
EXTRA="" run base X=1
EXTRA="--tools high" run high X=1
EXTRA="--tools high_b" run high_b X=1
EXTRA="--streams 8" run s8 X=1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zl/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h.get("cpu_ms_per_frame"), d["engine"]["pictures_per_batch"], d["roofline"]["frac"], (d.get("pcie_out") or {}).get("frac"))
PY
