# round 3, last call: the GPU suite, smoke(), a random sweep (small and 720p-class pictures), then the host-bound lines again (the B-picture parser changed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/f3
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/f3/gputests.log 2>&1; tail -3 gpurun_out/f3/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python tools/gpu_sweep.py 100 31 > gpurun_out/f3/sweep.log 2>&1; tail -2 gpurun_out/f3/sweep.log
timeout 900 python tools/gpu_sweep.py 30 32 big > gpurun_out/f3/sweep_big.log 2>&1; tail -2 gpurun_out/f3/sweep_big.log
timeout 300 python bench.py > gpurun_out/f3/r03_bench.json 2>/dev/null
timeout 300 python bench.py --tools high --no-cpu-baseline --no-single > gpurun_out/f3/r03_bench_high.json 2>/dev/null
timeout 300 python bench.py --tools high_b --no-cpu-baseline --no-single > gpurun_out/f3/r03_bench_high_b.json 2>/dev/null
timeout 300 python bench.py --tools paff_b --no-cpu-baseline --no-single > gpurun_out/f3/r03_bench_paff_b.json 2>/dev/null
timeout 300 python bench.py --tools high_b --width 3840 --height 2160 --frames 16 --steps 3 --no-cpu-baseline --no-single > gpurun_out/f3/r03_c2_4k.json 2>/dev/null
timeout 300 python bench.py --codec hevc --streams 16 --frames 32 --steps 3 --no-cpu-baseline --no-single > gpurun_out/f3/r03_hevc_bench_1920x1080.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/f3/r03_*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(f, 'unreadable'); continue
    print(os.path.basename(f), d['value'], d.get('scaling_bound'), d['host_cpu']['cpu_ms_per_frame'], d['host_cpu']['cpus_busy'], d['bit_exact'])
PY
