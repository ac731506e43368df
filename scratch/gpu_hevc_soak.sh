cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do for s in 16 6; do python bench.py --codec hevc --streams $s --frames 32 --steps 2 --no-cpu-baseline --no-single 2>&1 | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
for x in t:
    if x.startswith('bench.py'): print(x[:200])
l=json.loads(t[-1]); print('$s streams', l['value'], 'bit_exact', l['bit_exact'], l['engine']['pictures_per_batch'], 'errors', l['decode_errors'])"; done; done
