cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zh
run() { name=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2zh/$name.json 2>gpurun_out/r2zh/$name.err || echo "FAIL $name"; }
EXTRA="--tools high" run high X=1
EXTRA="--tools high_b" run high_b X=1
EXTRA="--tools high --device-output" run high_dev X=1
EXTRA="--tools high_b --device-output" run high_b_dev X=1
EXTRA="--codec hevc" run hevc X=1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zh/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h.get("cpu_ms_per_frame"), d["engine"]["pictures_per_batch"])
    for k,v in d["kernels"].items(): print("     ",k,v)
    print("     ", d.get("host_ms_per_picture"))
PY
