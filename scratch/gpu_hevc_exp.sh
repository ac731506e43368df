# timing experiment: k_hevc_intra with parts switched off (JM_AMD_DEC_EXP_HEVC bits: 1 no block loop, 2 no waiting on the row above, 4 no tile load / store)
cd $GRAFT_REPO_ROOT
run() { python bench.py --codec hevc --streams 4 --frames 32 --steps 2 --no-cpu-baseline --no-single 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['kernels']
print('$1', 'fps', l['value'], {a:(b['avg_us'],b['pictures_per_launch']) for a,b in k.items() if b['launches']})"; }
run base
for v in 1 2 4 3 5 7; do JM_AMD_DEC_EXP_HEVC=$v run exp$v; done
