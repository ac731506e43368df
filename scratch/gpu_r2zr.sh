cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zr
for s in 1 4 8 16; do timeout 300 python bench.py --codec hevc --streams $s --no-cpu-baseline --no-single --steps 3 > gpurun_out/r2zr/hevc_s$s.json 2>/dev/null || echo FAIL $s; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zr/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]; k=d["kernels"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], h["cpus_busy"], h["cpu_ms_per_frame"], d["engine"]["pictures_per_batch"], {n:(v["avg_us"], v["pictures_per_launch"]) for n,v in k.items() if v["launches"]})
PY
