# timing experiment: the deblocking band kernel with parts switched off (JM_DBG_DEBLOCK bits: 1 no chroma workgroups, 2 no luma workgroups, 4 no per-step
# barrier, 8 no ring hand-over stores / publishing, 16 no waiting for the band above, 32 no final stores, 64 no luma filter arithmetic)
cd $GRAFT_REPO_ROOT
run() { python bench.py --streams 16 --steps 3 --no-cpu-baseline --no-single --device-output 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['kernels']
print('$1', 'fps', l['value'], 'deblock', k['k_deblock']['avg_us'], k['k_deblock']['pictures_per_launch'], 'intra', k['k_intra']['avg_us'])"; }
export JM_AMD_DEC_CHAIN_DEPTH=1
run base
for v in 1 2 5 6 24 25 33 65 57; do JM_AMD_DEC_LIB=$PWD/jmcodec_amd/lib_dbg$v/libjm_amd_dec.so run dbg$v; done
