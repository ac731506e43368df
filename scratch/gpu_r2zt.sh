cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain or c4_slice or concurrent or recover or damaged or direct or c2_full" > gpurun_out/r2zt/pytest.txt 2>&1; tail -3 gpurun_out/r2zt/pytest.txt
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2zt/$name.json 2>gpurun_out/r2zt/$name.err || echo "FAIL $name"; }
for s in 20 24 32 48; do
EXTRA="--streams $s" run s${s}_split X=1
EXTRA="--streams $s" run s${s}_nosplit JM_AMD_DEC_LANE_SPLIT=0
done
EXTRA="--device-output" run dev_split X=1
EXTRA="--device-output" run dev_nosplit JM_AMD_DEC_LANE_SPLIT=0
EXTRA="--streams 16" run s16 X=1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zt/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h["cpu_ms_per_frame"], d["engine"]["pictures_per_batch"], d["engine"]["device_wait_errors"], d["roofline"]["frac"])
PY
