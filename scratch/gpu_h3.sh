set -x
timeout 900 python -m pytest tests/test_hevc_gpu_parity.py -x -q 2>&1 | tail -4
timeout 300 python bench.py --codec hevc --streams 8 --frames 32 --width 1920 --height 1080 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_hevc_1080p.json 2> gpurun_out/bench_hevc_1080p.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_hevc_1080p.json')); print(d['value'], d['host_ms_per_picture'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items()})"
timeout 300 python bench.py --codec hevc --streams 8 --frames 16 --width 3840 --height 2160 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_hevc_4k.json 2> gpurun_out/bench_hevc_4k.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_hevc_4k.json')); print(d['value'], d['host_ms_per_picture'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items()})"
