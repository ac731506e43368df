# round 4: the FIRST give-up of a chain launch (record behind the abort word): head / margin (12 more keys between the pictures of a chain) / lag2 at 4 and 8
# streams, JM_AMD_DEC_VERBOSE, until something gives up
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab6
python bench.py --no-extra --no-cpu-baseline --no-single --steps 3 > /dev/null 2>&1
for i in 1 2 3 4 5; do
  for w in head margin lag2; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so
    [ $w != head ] && L=$GRAFT_REPO_ROOT/jmcodec_amd/lib_dbg_$w/libjm_amd_dec.so
    for s in 4 8; do
      JM_AMD_DEC_VERBOSE=1 JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams $s > gpurun_out/ab6/${w}_s${s}_$i.json 2> gpurun_out/ab6/${w}_s${s}_$i.err
    done
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab6/*_s*_*.json")):
    try: l=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(f, "no line"); continue
    print(f.split("/")[-1], l["value"], "recoveries", l["engine"]["chain_recoveries_whole_run"], "chain batches", l["engine"]["chain_batches_whole_run"], l["kernels"]["k_chain"]["avg_us"])
PY
grep -h -B1 -A3 "FIRST give-up" gpurun_out/ab6/*.err | cut -c1-700 | head -60
