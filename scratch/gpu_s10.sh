for h in 0 1500 2500 4000; do
JM_AMD_DEC_INTRA_HOLD_US=$h timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('intra hold $h', d['value'], d['engine']['pictures_per_batch'], d['pcie_out']['achieved'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items()})"
done
