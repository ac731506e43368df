timeout 900 python -m pytest tests/test_hevc_gpu_parity.py -x -q 2>&1 | tail -2
for i in 1 2 3; do timeout 900 python -m pytest tests/test_hevc_gpu_parity.py -x -q -k "c3 or concurrent or b_gop8 or rps" 2>&1 | tail -1; done
timeout 300 python bench.py --codec hevc --streams 16 --frames 32 --width 1920 --height 1080 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('1080p', d['value'], d['host_ms_per_picture'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items()})"
timeout 300 python bench.py --codec hevc --streams 16 --frames 16 --width 3840 --height 2160 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('4k', d['value'], d['host_ms_per_picture'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items()})"
