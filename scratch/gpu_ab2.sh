cd $GRAFT_REPO_ROOT
run() { python bench.py --streams 1 --steps 6 --no-cpu-baseline --no-single 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['kernels']
print('$1', 'fps', l['value'], {a:(b['avg_us'],b['pictures_per_launch'],b['launches']) for a,b in k.items()}, 'batches', l['engine']['batches'], l['engine']['pictures_per_batch'], l['engine']['engine_thread_ms'], l['host_ms_per_picture'], 'wait', l['engine']['direct_output']['caller_wait_us_per_frame'])"; }
for rep in 1 2; do
JM_AMD_DEC_LIB=$PWD/jmcodec_amd/lib_old/libjm_amd_dec.so run old
run new
done
