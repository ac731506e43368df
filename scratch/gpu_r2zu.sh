cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zu
run() { name=$1; shift; timeout 200 env "$@" python bench.py --no-cpu-baseline --no-single $EXTRA > gpurun_out/r2zu/$name.json 2>gpurun_out/r2zu/$name.err || echo "FAIL $name"; }
for s in 6 8 12 16; do for cs in 4 8 16; do
EXTRA="--streams $s" run s${s}_cs$cs JM_AMD_DEC_CHAIN_STREAMS=$cs
done; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zu/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], d["engine"]["pictures_per_batch"], d["engine"]["chain_pictures_whole_run"])
PY
