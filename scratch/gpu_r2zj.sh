# two lanes of chain launches at few streams: parity + rates
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zj
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain or c4_slice or concurrent or recover or damaged" > gpurun_out/r2zj/pytest.txt 2>&1; tail -3 gpurun_out/r2zj/pytest.txt
run() { name=$1; shift; timeout 120 env "$@" python bench.py --no-cpu-baseline --no-single --steps 10 $EXTRA > gpurun_out/r2zj/$name.json 2>gpurun_out/r2zj/$name.err || echo "FAIL $name"; }
for s in 2 4 8 12 16; do
EXTRA="--streams $s" run s${s}_split JM_AMD_DEC_CHAIN_SPLIT=1
EXTRA="--streams $s" run s${s}_nosplit JM_AMD_DEC_CHAIN_SPLIT=0
done
EXTRA="--streams 32" run s32_split X=1
EXTRA="--streams 1" run s1_split X=1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zj/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], d["engine"]["pictures_per_batch"], d["engine"]["device_wait_errors"], d["kernels"]["k_chain"])
PY
