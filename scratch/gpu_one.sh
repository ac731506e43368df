cd $GRAFT_REPO_ROOT
JM_AMD_DEC_VERBOSE=1 python bench.py --streams 1 --steps 4 --no-cpu-baseline --no-single 2> gpurun_out/one.err | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['kernels']
print('fps', l['value'], l['engine'], {a:(b['avg_us'],b['pictures_per_launch'],b['launches']) for a,b in k.items()})"
grep -v "amdgpu.ids" gpurun_out/one.err | sort | uniq -c | sort -rn | head -12
