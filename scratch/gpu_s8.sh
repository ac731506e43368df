for f in 60 120 240; do
timeout 400 python bench.py --frames $f --steps 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('frames $f', d['value'], d['engine']['pictures_per_batch'], d['ms_per_step'])"
done
