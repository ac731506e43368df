# D2H mechanisms: SDMA (one engine per direction) against the runtime's blit kernels (HSA_ENABLE_SDMA=0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2z4
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2z4/$name.json 2>gpurun_out/r2z4/$name.err || echo "FAIL $name"; }
EXTRA="" run a_default X=1
EXTRA="" run b_default_nosdma HSA_ENABLE_SDMA=0
EXTRA="" run c_direct_nosdma HSA_ENABLE_SDMA=0 JM_AMD_DEC_OUT_FETCH=direct
EXTRA="" run d_pinned_nosdma HSA_ENABLE_SDMA=0 JM_AMD_DEC_OUT_PINNED=1
EXTRA="" run e_fetchall_nosdma HSA_ENABLE_SDMA=0 JM_AMD_DEC_OUT_FETCH=1/1
EXTRA="" run f_fetchall X=1 JM_AMD_DEC_OUT_FETCH=1/1
EXTRA="--streams 8" run g_s8_direct_nosdma HSA_ENABLE_SDMA=0 JM_AMD_DEC_OUT_FETCH=direct
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2z4/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h.get("cpu_ms_per_frame"), h["calling_threads"]["cpu_ms_per_frame"], d["engine"]["pictures_per_batch"])
PY
