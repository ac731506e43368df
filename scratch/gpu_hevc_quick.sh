cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/h3
timeout 900 python -m pytest tests/test_hevc_gpu_parity.py -m gpu -x -q -k "bit_exact" > gpurun_out/h3/gputests.log 2>&1; tail -2 gpurun_out/h3/gputests.log
for a in "--streams 1" "--streams 16"; do
timeout 400 python bench.py --codec hevc $a --frames 32 --steps 3 --no-cpu-baseline > gpurun_out/h3/bench.json 2> gpurun_out/h3/bench.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/h3/bench.json").read().strip().splitlines()[-1])
print("value", l["value"], "bit_exact", l["bit_exact"], "single", l.get("single_stream",{}).get("value"), {k:(v["avg_us"],v["pictures_per_launch"]) for k,v in l["kernels"].items() if k in ("k_intra","k_inter")})
PY
done
