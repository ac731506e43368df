set -x
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_pad.json 2> gpurun_out/bench_pad.err
tail -c 1500 gpurun_out/bench_pad.json
