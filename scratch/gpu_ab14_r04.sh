# round 4: flt_luma returns at once when no lane of the wave has a boundary strength on the edge (head) against the library before it (scratch/_ab/base_*):
# whole GPU suite on head, then 1 / 4 / 8 streams and the device-resident default workload
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; P=gpurun_out/ab14; mkdir -p $P
OLD=$GRAFT_REPO_ROOT/scratch/_ab/base_libjm_amd_dec.so
timeout 1500 python -m pytest tests -m gpu -x -q -rs > $P/gputests.log 2>&1; tail -4 $P/gputests.log
for i in 1 2 3; do
  for w in head base; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = base ] && L=$OLD
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --device-output > $P/${w}_dev_$i.json 2> $P/${w}_dev_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 8 > $P/${w}_s8_$i.json 2> $P/${w}_s8_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 4 > $P/${w}_s4_$i.json 2> $P/${w}_s4_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 1 > $P/${w}_s1_$i.json 2> $P/${w}_s1_$i.err
  done
done
python tools/ab_summary.py $P > $P/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab14/summary.json"))
for k,v in sorted(d.items()):
    print(k, v["value"], {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in v["kernels"].items()})
PY
