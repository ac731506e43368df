# where is the 32-stream bound now that the host has CPU headroom?  output routes, stream counts, batches in flight
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2z2
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2z2/$name.json 2>/dev/null || echo "FAIL $name"; }
EXTRA="" run a_default_1 X=1
EXTRA="" run b_pinned JM_AMD_DEC_OUT_PINNED=1
EXTRA="" run c_fetch_1_4 JM_AMD_DEC_OUT_FETCH=1/4
EXTRA="" run d_fetch_1_2 JM_AMD_DEC_OUT_FETCH=1/2
EXTRA="" run e_fetch_auto JM_AMD_DEC_OUT_FETCH=auto
EXTRA="--streams 48" run f_s48 X=1
EXTRA="--streams 64" run g_s64 X=1
EXTRA="" run h_inflight3 JM_AMD_DEC_INFLIGHT=3
EXTRA="" run i_default_2 X=1
EXTRA="--streams 48" run j_s48_pinned JM_AMD_DEC_OUT_PINNED=1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2z2/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h.get("cpu_ms_per_frame"), h["calling_threads"]["cpu_ms_per_frame"], d["engine"].get("pictures_per_launch"), d.get("kernels",{}).get("k_deblock_band",{}))
PY
