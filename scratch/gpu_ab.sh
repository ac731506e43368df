# A/B on one box: jmcodec_amd/lib_old (the previous build) against jmcodec_amd/lib, alternating, at 1 / 8 / 32 streams
cd $GRAFT_REPO_ROOT
run() { python bench.py $2 --steps 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['kernels']
print('$1', '$2', 'fps', l['value'], 'single', l.get('single_stream',{}).get('value'), 'dev', l.get('device_resident_output',{}).get('value'), 'deblock', k['k_deblock']['avg_us'], k['k_deblock']['pictures_per_launch'], 'chain', k['k_chain']['avg_us'], k['k_chain']['pictures_per_launch'], 'intra', k['k_intra']['avg_us'], 'cpu', l['host_cpu']['cpu_ms_per_frame'])"; }
for rep in 1 2; do for s in "--streams 1" "--streams 8" ""; do
JM_AMD_DEC_LIB=$PWD/jmcodec_amd/lib_old/libjm_amd_dec.so run old "$s"
run new "$s"
done; done
