# round 4: why do chain launches of 4 / 8 streams give up with the third kernel set?  Variants at 4 and 8 streams, each with JM_AMD_DEC_VERBOSE (dump of
# the band counters at give-up; the census is off): head (extra spacing behind intra-role pictures), nofix (the library that failed), pad (3 workgroups
# per CU), lag2 (two rows per deblocking step), v1 (previous commit).  Then the deblocking variants d2 / d4 / p1 on the default workload.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab5
python bench.py --no-extra --no-cpu-baseline --no-single --steps 3 > /dev/null 2>&1
for i in 1 2 3; do
  for w in head nofix pad lag2 v1; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so
    [ $w = v1 ] && L=$GRAFT_REPO_ROOT/scratch/_ab/v1/jmcodec_amd/lib/libjm_amd_dec.so
    [ $w = nofix ] && L=$GRAFT_REPO_ROOT/jmcodec_amd/lib_dbg_nofix/libjm_amd_dec.so
    [ $w = pad ] && L=$GRAFT_REPO_ROOT/jmcodec_amd/lib_dbg_pad/libjm_amd_dec.so
    [ $w = lag2 ] && L=$GRAFT_REPO_ROOT/jmcodec_amd/lib_dbg_lag2/libjm_amd_dec.so
    for s in 4 8; do
      JM_AMD_DEC_VERBOSE=1 JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams $s > gpurun_out/ab5/${w}_s${s}_$i.json 2> gpurun_out/ab5/${w}_s${s}_$i.err
    done
  done
done
python tools/ab_summary.py gpurun_out/ab5 > gpurun_out/ab5/summary.json
python - <<'PY'
import json,glob
d=json.load(open("gpurun_out/ab5/summary.json"))
for k,v in sorted(d.items()):
    print(k, v["value"], {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in v["kernels"].items() if kk in ("k_chain","k_inter")})
for f in sorted(glob.glob("gpurun_out/ab5/*_s*_*.json")):
    try: l=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: continue
    if l["engine"]["chain_recoveries_whole_run"]: print(f.split("/")[-1], "recoveries", l["engine"]["chain_recoveries_whole_run"], "chain batches", l["engine"]["chain_batches_whole_run"])
PY
for f in gpurun_out/ab5/nofix_s8_1.err gpurun_out/ab5/nofix_s4_1.err gpurun_out/ab5/head_s8_1.err; do echo "== $f"; grep -A34 "gave up" $f | cut -c1-900 | head -40; done
for i in 1 2; do
  for w in head d2 d4 p1; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so
    [ $w != head ] && L=$GRAFT_REPO_ROOT/jmcodec_amd/lib_dbg_$w/libjm_amd_dec.so
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --device-output > gpurun_out/ab5/${w}_dev_$i.json 2> gpurun_out/ab5/${w}_dev_$i.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab5/*_dev_*.json")):
    try: l=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: continue
    print(f.split("/")[-1], l["value"], {k:(v["avg_us"],v["pictures_per_launch"]) for k,v in l["kernels"].items() if k in ("k_deblock",)})
PY
