for s in 32 64; do
timeout 300 python bench.py --streams $s --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $s', d['value'], d['engine'], d['ms_per_step'])"
done
