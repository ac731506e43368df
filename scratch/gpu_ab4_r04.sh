# round 4: chain kernels without the per-workgroup census (it is diagnostic-only now) against the previous commit (v1) at 1 / 8 / 16 streams, then the
# chain diagnostics (scratch/gpu_chain_r04.sh: C2 with chains forced on, census + dump; SQ / TCC counters of k_chain at 8 streams)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab4
python bench.py --no-extra --no-cpu-baseline --no-single --steps 3 > /dev/null 2>&1
for i in 1 2; do
  for w in v1 head; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so
    [ $w = v1 ] && L=$GRAFT_REPO_ROOT/scratch/_ab/v1/jmcodec_amd/lib/libjm_amd_dec.so
    for s in 1 2 4 8 16; do
      JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams $s > gpurun_out/ab4/${w}_s${s}_$i.json 2> gpurun_out/ab4/${w}_s${s}_$i.err
    done
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 > gpurun_out/ab4/${w}_host_$i.json 2> gpurun_out/ab4/${w}_host_$i.err
  done
done
python tools/ab_summary.py gpurun_out/ab4 > gpurun_out/ab4/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab4/summary.json"))
for k,v in sorted(d.items()):
    print(k, v["value"], {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in v["kernels"].items()})
PY
bash scratch/gpu_chain_r04.sh
