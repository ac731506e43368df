# round 5, step 1b: packed-math k_recon_inter -- the WHOLE GPU suite on head, SQ counters of head (C1, C2), then head against the round-4 library
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; P=gpurun_out/ab2; mkdir -p $P
R4=$GRAFT_REPO_ROOT/scratch/_ab/r4/libjm_amd_dec.so
timeout 1500 python -m pytest tests -m gpu -x -q -rs > $P/gputests.log 2>&1; tail -6 $P/gputests.log
bash scratch/gpu_sq_r05.sh head jmcodec_amd/lib/libjm_amd_dec.so c1 c2 > $P/sq.log 2>&1; grep recon_inter $P/sq.log
for i in 1 2; do
  for w in head r4; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = r4 ] && L=$R4
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --device-output > $P/${w}_dev_$i.json 2> $P/${w}_dev_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 8 > $P/${w}_s8_$i.json 2> $P/${w}_s8_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 1 > $P/${w}_s1_$i.json 2> $P/${w}_s1_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --device-output > $P/${w}_c2_$i.json 2> $P/${w}_c2_$i.err
  done
done
python tools/ab_summary.py $P > $P/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab2/summary.json"))
for k,v in sorted(d.items()):
    print(k, v["value"], {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in v["kernels"].items()})
PY
