for s in 32 64; do
JM_AMD_DEC_EXP_NOPACK=1 timeout 300 python bench.py --streams $s --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NOPACK streams $s', d['value'], d['engine'], d['host_ms_per_picture'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items() if k!='k_intra'})"
done
