set -x
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python bench.py --codec hevc --streams 8 --frames 32 --width 1920 --height 1080 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_hevc_1080p.json 2> gpurun_out/bench_hevc_1080p.err; tail -c 1800 gpurun_out/bench_hevc_1080p.json
timeout 300 python bench.py --codec hevc --streams 8 --frames 16 --width 3840 --height 2160 --steps 2 --warmup 1 > gpurun_out/bench_hevc_4k.json 2> gpurun_out/bench_hevc_4k.err; tail -c 2200 gpurun_out/bench_hevc_4k.json
