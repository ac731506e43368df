# round 3: PAFF on the device: parity cases, full size, lone field, mixed batches, smoke(), a random sweep with field pictures, the two PAFF bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/paff
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "paff or partner or field" > gpurun_out/paff/t1.log 2>&1; tail -5 gpurun_out/paff/t1.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke OK"
timeout 1200 python tools/gpu_sweep.py ${1:-60} 37 > gpurun_out/paff/sweep.log 2>&1; tail -3 gpurun_out/paff/sweep.log
for t in paff paff_b; do
timeout 600 python bench.py --tools $t --no-cpu-baseline --no-single > gpurun_out/paff/r03_bench_$t.json 2> gpurun_out/paff/bench_$t.err; tail -2 gpurun_out/paff/bench_$t.err | grep -v amdgpu.ids
python - $t <<'PY'
import json, sys
l=json.loads(open(f"gpurun_out/paff/r03_bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "value", l["value"], "bit_exact", l["bit_exact"], l["frames_checked"], "bound", l.get("scaling_bound"), "cpu_ms", l["host_cpu"]["cpu_ms_per_frame"])
print({k:(v["avg_us"],v["pictures_per_launch"]) for k,v in l["kernels"].items()})
PY
done
