for s in 32 48 64 96; do
timeout 300 python bench.py --streams $s --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $s', d['value'], d['pcie_out']['achieved'], d['host_ms_per_picture'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items()})"
done
