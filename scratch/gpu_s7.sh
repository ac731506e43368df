for h in 0 400 700 1000 1500; do
JM_AMD_DEC_HOLD_US=$h timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('hold $h', d['value'], d['engine']['pictures_per_batch'], d['host_ms_per_picture']['wait_slot_ns'], {k:(v['avg_us']) for k,v in d['kernels'].items()})"
done
