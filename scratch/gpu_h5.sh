for e in 0 1 2 4 7; do
JM_AMD_DEC_EXP_HEVC=$e timeout 300 python bench.py --codec hevc --streams 8 --frames 16 --width 3840 --height 2160 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('exp $e', d['value'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items()})"
done
