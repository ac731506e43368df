cd $GRAFT_REPO_ROOT
run() { python bench.py --streams 1 --steps 6 --no-cpu-baseline --no-single 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['kernels']
print('$1', 'fps', l['value'], 'chain', k['k_chain']['avg_us'], k['k_chain']['pictures_per_launch'], 'batches', l['engine']['batches'], l['host_ms_per_picture'], l['host_memory']['job_slots_mb'], l['host_memory']['job_slots_grown'])"; }
for rep in 1 2; do
JM_AMD_DEC_LIB=$PWD/jmcodec_amd/lib_old/libjm_amd_dec.so run old
run new
JM_AMD_DEC_JOB_WORST_CASE=1 run new_worstcase_slots
JM_AMD_DEC_LIB=$PWD/jmcodec_amd/lib_old/libjm_amd_dec.so JM_AMD_DEC_JOB_WORST_CASE=1 run old_worstcase_slots
done
