# round 4: the one-window ('uni') reconstruction path -- suite on head, then head against the previous commit's library (scratch/_ab/prev) device-resident
# default workload, 8 / 1 streams, and the C1 traffic passes (FETCH_SIZE / WRITE_SIZE, stage kernels and chain launches) of head
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; P=gpurun_out/ab9; mkdir -p $P
PREV=$GRAFT_REPO_ROOT/scratch/_ab/prev/jmcodec_amd/lib/libjm_amd_dec.so
timeout 1500 python -m pytest tests -m gpu -x -q -rs > $P/gputests.log 2>&1; tail -4 $P/gputests.log
for i in 1 2 3; do
  for w in head prev; do
    L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = prev ] && L=$PREV
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --device-output > $P/${w}_dev_$i.json 2> $P/${w}_dev_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 8 > $P/${w}_s8_$i.json 2> $P/${w}_s8_$i.err
    JM_AMD_DEC_LIB=$L timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams 1 > $P/${w}_s1_$i.json 2> $P/${w}_s1_$i.err
  done
done
python tools/ab_summary.py $P > $P/summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab9/summary.json"))
for k,v in sorted(d.items()):
    print(k, v["value"], {kk:(vv["avg_us"],vv["pictures_per_launch"]) for kk,vv in v["kernels"].items()})
PY
for w in head prev; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = prev ] && L=$PREV
  export JM_AMD_DEC_LIB=$L
  for c in FETCH_SIZE WRITE_SIZE; do
    JM_AMD_DEC_CHAIN_DEPTH=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --frames 30 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-single --no-extra --device-output > /dev/null 2>&1
    find gpurun_out/prof_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $P/pmc_${w}_c1_$c.csv; rm -rf gpurun_out/prof_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --steps 1 --warmup 1 --frames 30 --streams 1 --no-cpu-baseline --no-single --no-extra --device-output > $P/pmc_${w}_c1chain_line_$c.json 2>/dev/null
    find gpurun_out/prof_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $P/pmc_${w}_c1chain_$c.csv; rm -rf gpurun_out/prof_$c
  done
  python3 tools/make_traffic_profile.py --fetch $P/pmc_${w}_c1_FETCH_SIZE.csv --write $P/pmc_${w}_c1_WRITE_SIZE.csv --out $P/${w}_pmc_traffic_h264_baseline_1920x1080.json --width 1920 --height 1080 \
    --env "JM_AMD_DEC_CHAIN_DEPTH=1" --command "python3 bench.py --frames 30 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-single --device-output" \
    --chain-fetch $P/pmc_${w}_c1chain_FETCH_SIZE.csv --chain-write $P/pmc_${w}_c1chain_WRITE_SIZE.csv --chain-line $P/pmc_${w}_c1chain_line_FETCH_SIZE.json > $P/pmc_summary_${w}.txt 2>&1
  cat $P/pmc_summary_${w}.txt | tail -12
done
unset JM_AMD_DEC_LIB
rm -f $P/pmc_*.csv
