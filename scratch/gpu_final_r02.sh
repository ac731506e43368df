# scratch/gpu_final_r02.sh -- everything the round's final numbers come from, in one gpurun call: GPU suite, smoke, profile set, other configs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.txt 2>&1; tail -2 gpurun_out/final/pytest.txt
timeout 300 python __graft_entry__.py smoke > gpurun_out/final/smoke.txt 2>&1; tail -3 gpurun_out/final/smoke.txt
bash scratch/gpu_prof_r02.sh > gpurun_out/final/prof_log.txt 2>&1
timeout 500 python bench.py --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-cpu-baseline --no-single > gpurun_out/p2/r02_c2_4k.json 2>/dev/null || echo FAIL c2
timeout 500 python bench.py --codec hevc --streams 16 --frames 32 --width 1920 --height 1080 --steps 3 --no-cpu-baseline --no-single > gpurun_out/p2/r02_hevc_bench_1080p.json 2>/dev/null || echo FAIL h1
timeout 500 python bench.py --codec hevc --streams 16 --frames 16 --width 3840 --height 2160 --steps 3 --no-cpu-baseline --no-single > gpurun_out/p2/r02_hevc_bench_4k.json 2>/dev/null || echo FAIL h4
timeout 300 python bench.py --tools high --no-cpu-baseline --no-single > gpurun_out/p2/r02_bench_high.json 2>/dev/null || echo FAIL high
timeout 300 python bench.py --tools high_b --no-cpu-baseline --no-single > gpurun_out/p2/r02_bench_high_b.json 2>/dev/null || echo FAIL high_b
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/p2/r02_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: continue
    if "value" not in d: continue
    h=d["host_cpu"]; r=d["roofline"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h["cpu_ms_per_frame"], r["kernel"], r["frac"], (d.get("pcie_out") or {}).get("frac"), (d.get("single_stream") or {}).get("value"))
PY
