# BASELINE configs 2 and 3 at their sizes (diagnostic lines for profiles/)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zm
timeout 500 python bench.py --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-cpu-baseline --no-single > gpurun_out/r2zm/r02_c2_4k.json 2>gpurun_out/r2zm/c2.err || echo FAIL c2
timeout 500 python bench.py --codec hevc --streams 16 --frames 32 --width 1920 --height 1080 --steps 3 --no-cpu-baseline --no-single > gpurun_out/r2zm/r02_hevc_bench_1080p.json 2>gpurun_out/r2zm/h1.err || echo FAIL h1
timeout 500 python bench.py --codec hevc --streams 16 --frames 16 --width 3840 --height 2160 --steps 3 --no-cpu-baseline --no-single > gpurun_out/r2zm/r02_hevc_bench_4k.json 2>gpurun_out/r2zm/h4.err || echo FAIL h4
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zm/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h.get("cpu_ms_per_frame"), d["engine"]["pictures_per_batch"], (d.get("pcie_out") or {}).get("achieved"))
PY
tail -2 gpurun_out/r2zm/*.err
