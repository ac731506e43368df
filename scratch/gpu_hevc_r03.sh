# round 3: HEVC GPU parity + HEVC bench at 1 and 16 streams (1080p)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/h3
timeout 1500 python -m pytest tests/test_hevc_gpu_parity.py -m gpu -x -q > gpurun_out/h3/gputests.log 2>&1; tail -4 gpurun_out/h3/gputests.log
for a in "--streams 1" "--streams 4" "--streams 16"; do
timeout 400 python bench.py --codec hevc $a --frames 32 --steps 3 --no-cpu-baseline > gpurun_out/h3/bench.json 2> gpurun_out/h3/bench.err; tail -2 gpurun_out/h3/bench.err | grep -v amdgpu.ids
python - <<'PY'
import json
l=json.loads(open("gpurun_out/h3/bench.json").read().strip().splitlines()[-1])
print("value", l["value"], "bit_exact", l["bit_exact"], "cpu_ms", l["host_cpu"]["cpu_ms_per_frame"], "busy", l["host_cpu"]["cpus_busy"], "single", l.get("single_stream",{}).get("value"), "batches", l["engine"]["batches"], l["engine"]["pictures_per_batch"])
print({k:(v["avg_us"],v["pictures_per_launch"],v["launches"]) for k,v in l["kernels"].items()})
PY
done
