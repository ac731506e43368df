for st in 0 1 2 3; do
timeout 400 python bench.py --stagger-ms $st --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stagger $st', d['value'], d['engine']['pictures_per_batch'], d['ms_per_step'], d['pcie_out']['achieved'])"
done
