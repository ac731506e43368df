# round 4: the bands of a picture reconstructed before the chain launch (PS_RECON) fetch their samples through the caches (head) against the library
# before it (scratch/_ab/noprio_*): chain tests on head, then 1 / 4 / 8 streams, device-resident default workload, a one-stream time line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; P=gpurun_out/ab13; mkdir -p $P
OLD=$GRAFT_REPO_ROOT/scratch/_ab/noprio_libjm_amd_dec.so
timeout 900 python -m pytest tests -m gpu -x -q -rs -k "chain or stall or recover" > $P/gputests_chain.log 2>&1; tail -4 $P/gputests_chain.log
for i in 1 2 3; do
  for w in head noprio; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = noprio ] && L=$OLD
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --device-output > $P/${w}_dev_$i.json 2> $P/${w}_dev_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 8 > $P/${w}_s8_$i.json 2> $P/${w}_s8_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 4 > $P/${w}_s4_$i.json 2> $P/${w}_s4_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 1 > $P/${w}_s1_$i.json 2> $P/${w}_s1_$i.err
  done
done
python tools/ab_summary.py $P > $P/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab13/summary.json"))
for k,v in sorted(d.items()):
    print(k, v["value"], {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in v["kernels"].items()})
PY
L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so
JM_AMD_DEC_CENSUS=1 JM_AMD_DEC_CHAIN_TIMELINE=1 timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 2 --warmup 1 --streams 1 > /dev/null 2> $P/tl_head.err
awk '/chain launch of 8/{n++} n==3' $P/tl_head.err | grep -E "time line|reconstruction workgroups" | cut -c1-160
