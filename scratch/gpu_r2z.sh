# host-parser fast path: parity suite + the bench at several stream counts
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2z
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2z/pytest.txt 2>&1; tail -3 gpurun_out/r2z/pytest.txt
for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --no-single > gpurun_out/r2z/s32_$i.json 2>/dev/null || echo "FAIL s32 $i"; done
for s in 1 8 16; do timeout 120 python bench.py --streams $s --no-cpu-baseline --no-single > gpurun_out/r2z/s${s}.json 2>/dev/null || echo "FAIL s$s"; done
timeout 200 python bench.py --no-cpu-baseline --no-single --device-output > gpurun_out/r2z/s32_dev.json 2>/dev/null || echo "FAIL dev"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2z/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], d["engine"]["device_wait_errors"], h["cpus_busy"], h.get("cpu_ms_per_frame"), h.get("calling_threads"), h.get("parse_ms_per_frame"))
PY
