# what one of eight ranks gets under a 16-CPU allotment: the bench pinned to 2 (and 4) CPUs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zq
for c in 2 4 8; do
  timeout 400 taskset -c 0-$((c-1)) python bench.py --no-cpu-baseline --no-single --steps 4 > gpurun_out/r2zq/cpus$c.json 2>gpurun_out/r2zq/cpus$c.err || echo FAIL $c
done
timeout 400 taskset -c 0-1 python bench.py --no-cpu-baseline --no-single --steps 4 --streams 8 > gpurun_out/r2zq/cpus2_s8.json 2>/dev/null || echo FAIL s8
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2zq/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    h=d["host_cpu"]
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], h["cpus_busy"], h["cpu_ms_per_frame"], d["config"]["host_parse_threads"], d["engine"]["pictures_per_batch"])
PY
