# round 4: wait_final naps in proportion to the steps still missing ('nap', head) against the library before it (scratch/_ab/uni): chain tests on head, then
# 1 / 4 / 8 streams, device-resident default workload, and the chain traffic passes (FETCH_SIZE / WRITE_SIZE) of both
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; P=gpurun_out/ab10; mkdir -p $P
OLD=$GRAFT_REPO_ROOT/scratch/_ab/uni/libjm_amd_dec.so
timeout 900 python -m pytest tests -m gpu -x -q -rs -k "chain or stall or recover" > $P/gputests_chain.log 2>&1; tail -4 $P/gputests_chain.log
for i in 1 2 3; do
  for w in head uni; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = uni ] && L=$OLD
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --device-output > $P/${w}_dev_$i.json 2> $P/${w}_dev_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 8 > $P/${w}_s8_$i.json 2> $P/${w}_s8_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 4 > $P/${w}_s4_$i.json 2> $P/${w}_s4_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 1 > $P/${w}_s1_$i.json 2> $P/${w}_s1_$i.err
  done
done
python tools/ab_summary.py $P > $P/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab10/summary.json"))
for k,v in sorted(d.items()):
    print(k, v["value"], {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in v["kernels"].items()})
PY
for w in head uni; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = uni ] && L=$OLD
  export JM_AMD_DEC_LIB=$L
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --steps 1 --warmup 1 --frames 30 --streams 1 --no-cpu-baseline --no-single --no-extra --device-output > $P/pmc_${w}_c1chain_line_$c.json 2>/dev/null
    find gpurun_out/prof_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $P/pmc_${w}_c1chain_$c.csv; rm -rf gpurun_out/prof_$c
  done
  for c in FETCH_SIZE WRITE_SIZE; do echo -n "$w "; python3 tools/pmc_summary.py $P/pmc_${w}_c1chain_$c.csv 2>&1 | grep -i chain | head -2; done
done
unset JM_AMD_DEC_LIB
rm -f $P/pmc_*.csv
