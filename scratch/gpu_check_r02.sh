# scratch/gpu_check_r02.sh -- last check of the round's build: GPU suite, smoke, a random sweep, the default bench line and the HEVC lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/check
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/check/pytest.txt 2>&1; tail -2 gpurun_out/check/pytest.txt
timeout 300 python __graft_entry__.py smoke > gpurun_out/check/smoke.txt 2>&1; tail -1 gpurun_out/check/smoke.txt
timeout 400 python tools/gpu_sweep.py 150 9100 > gpurun_out/check/sweep.txt 2>&1; tail -2 gpurun_out/check/sweep.txt
timeout 400 python bench.py > gpurun_out/check/bench.json 2>gpurun_out/check/bench.err || echo FAIL bench
timeout 300 python bench.py --codec hevc --streams 16 --frames 32 --width 1920 --height 1080 --steps 3 --no-cpu-baseline --no-single > gpurun_out/check/hevc_1080p.json 2>/dev/null || echo FAIL h1
timeout 300 python bench.py --codec hevc --streams 16 --frames 16 --width 3840 --height 2160 --steps 3 --no-cpu-baseline --no-single > gpurun_out/check/hevc_4k.json 2>/dev/null || echo FAIL h4
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/check/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]; r=d["roofline"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h["cpu_ms_per_frame"], r["kernel"], r["frac"], (d.get("pcie_out") or {}).get("frac"), (d.get("single_stream") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
PY
# the CABAC-bound configurations (host half rebuilt with the packed table rows and BMI2)
mkdir -p gpurun_out/check
timeout 300 python bench.py --tools high --no-cpu-baseline --no-single > gpurun_out/check/r02_bench_high.json 2>/dev/null || echo FAIL high
timeout 300 python bench.py --tools high_b --no-cpu-baseline --no-single > gpurun_out/check/r02_bench_high_b.json 2>/dev/null || echo FAIL high_b
timeout 500 python bench.py --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-cpu-baseline --no-single > gpurun_out/check/r02_c2_4k.json 2>/dev/null || echo FAIL c2
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/check/r02_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h["cpu_ms_per_frame"])
PY
