# round 4: chain launches with the key slope per picture: 4 / 8 streams x 8 runs each (JM_AMD_DEC_VERBOSE), 1 / 2 / 16 streams, the GPU suite, then the
# deblocking variants (depth 2, pub 1, both) at 8 streams and on the default workload with device-resident output
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab7
python bench.py --no-extra --no-cpu-baseline --no-single --steps 3 > /dev/null 2>&1
for i in 1 2 3 4 5 6 7 8; do
  for s in 4 8; do
    JM_AMD_DEC_VERBOSE=1 timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams $s > gpurun_out/ab7/head_s${s}_$i.json 2> gpurun_out/ab7/head_s${s}_$i.err
  done
done
for i in 1 2; do for s in 1 2 3 6 12 16; do
  JM_AMD_DEC_VERBOSE=1 timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams $s > gpurun_out/ab7/head_s${s}_$i.json 2> gpurun_out/ab7/head_s${s}_$i.err
done; done
timeout 1500 python -m pytest tests -m gpu -x -q -rs > gpurun_out/ab7/gputests.log 2>&1; tail -4 gpurun_out/ab7/gputests.log
for i in 1 2; do
  for w in head d2 p1 d2p1; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so
    [ $w != head ] && L=$GRAFT_REPO_ROOT/jmcodec_amd/lib_dbg_$w/libjm_amd_dec.so
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --device-output > gpurun_out/ab7/${w}_dev_$i.json 2> gpurun_out/ab7/${w}_dev_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 8 > gpurun_out/ab7/${w}_v8_$i.json 2> gpurun_out/ab7/${w}_v8_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 1 > gpurun_out/ab7/${w}_v1_$i.json 2> gpurun_out/ab7/${w}_v1_$i.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab7/*.json")):
    try: l=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(f, "no line"); continue
    print(f.split("/")[-1], l["value"], "recov", l["engine"]["chain_recoveries_whole_run"], "chain", l["engine"]["chain_batches_whole_run"], {k:(v["avg_us"],v["pictures_per_launch"]) for k,v in l["kernels"].items() if k in ("k_chain","k_deblock") and v["launches"]})
PY
grep -h "FIRST give-up" gpurun_out/ab7/*.err | cut -c1-400 | head
