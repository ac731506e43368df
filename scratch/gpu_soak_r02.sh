# scratch/gpu_soak_r02.sh -- repeated runs of the chained configurations: every line must be bit-exact with zero device wait errors
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/soak
for i in 1 2 3 4; do for s in 1 3 8 12 16 20; do timeout 120 python bench.py --streams $s --no-cpu-baseline --no-single > gpurun_out/soak/s${s}_$i.json 2>/dev/null || echo "FAIL s$s run $i rc=$?"; done; done
for i in 1 2 3; do timeout 200 python bench.py --no-cpu-baseline --no-single > gpurun_out/soak/s32_$i.json 2>/dev/null || echo "FAIL s32 run $i"; done
for s in 1 4 16; do timeout 200 python bench.py --codec hevc --streams $s --steps 3 --no-cpu-baseline --no-single > gpurun_out/soak/hevc_s$s.json 2>/dev/null || echo "FAIL hevc $s"; done
for t in high high_b; do timeout 120 python bench.py --streams 6 --tools $t --no-cpu-baseline --no-single > gpurun_out/soak/s6_$t.json 2>/dev/null || echo "FAIL $t"; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/soak/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(f.split("/")[-1], d["value"], d["bit_exact"], d["decode_errors"], d["engine"]["device_wait_errors"], d["host_cpu"]["cpus_busy"])
PY
