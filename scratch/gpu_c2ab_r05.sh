# round 5: two-list windows fetched before filtering -- the H.264 GPU suite, then C2 head against the previous library (scratch/_ab/prev) on one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; P=gpurun_out/c2ab; mkdir -p $P
PREV=$GRAFT_REPO_ROOT/scratch/_ab/prev/libjm_amd_dec.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $P/gputests.log 2>&1; tail -3 $P/gputests.log
for i in 1 2 3; do
  for w in head prev; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = prev ] && L=$PREV
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --device-output > $P/${w}_c2_$i.json 2> $P/${w}_c2_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --tools high_b --steps 6 --device-output > $P/${w}_hb1080_$i.json 2> $P/${w}_hb1080_$i.err
  done
done
python tools/ab_summary.py $P > $P/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/c2ab/summary.json"))
for k,v in sorted(d.items()):
    print(k, v["value"], {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in v["kernels"].items()})
PY
