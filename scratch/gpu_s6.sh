for s in 32 64; do
JM_AMD_DEC_EXP_NOUPLWAIT=1 JM_AMD_DEC_EXP_NOPACK=1 timeout 300 python bench.py --streams $s --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NOUPLWAIT+NOPACK streams $s', d['value'], d['engine'], d['host_ms_per_picture'])"
done
JM_AMD_DEC_EXP_NOUPLWAIT=1 timeout 300 python bench.py --streams 32 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NOUPLWAIT streams 32', d['value'], d['engine'], d['host_ms_per_picture'])"
