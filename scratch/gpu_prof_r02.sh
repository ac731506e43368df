# scratch/gpu_prof_r02.sh -- round 2 profile set (run on the GPU box: gpurun -- 'bash scratch/gpu_prof_r02.sh'); results under gpurun_out/p2/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=gpurun_out/p2; mkdir -p $P
# 1. default bench line (the driver's command)
timeout 300 python bench.py > $P/r02_bench.json 2> $P/r02_bench.err
# 2. device-resident output (no D2H copies): plain, then under rocprofv3 --kernel-trace --stats -- the CSV must reproduce the line's roofline
timeout 300 python bench.py --device-output --no-cpu-baseline --no-single > $P/r02_bench_device_output.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_a -- python3 bench.py --device-output --no-cpu-baseline --no-single > $P/r02_bench_device_output_under_rocprof.json 2>/dev/null
find gpurun_out/prof_a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $P/r02_kernel_stats.csv; rm -rf gpurun_out/prof_a
# 3. few streams (chain launches): 1 and 8 streams, plain and under rocprofv3
for s in 1 8; do
  timeout 300 python bench.py --streams $s --no-cpu-baseline > $P/r02_bench_s$s.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s -- python3 bench.py --streams $s --device-output --no-cpu-baseline --no-single > $P/r02_bench_s${s}_device_output_under_rocprof.json 2>/dev/null
  find gpurun_out/prof_s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $P/r02_kernel_stats_s$s.csv; rm -rf gpurun_out/prof_s
done
# 4. PMC traffic per picture: one stream; stage kernels (chain off: one picture per launch) and chain launches; FETCH_SIZE / WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  JM_AMD_DEC_CHAIN_DEPTH=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --steps 1 --warmup 1 --frames 30 --streams 1 --no-cpu-baseline --no-single --device-output > /dev/null 2>&1
  find gpurun_out/prof_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $P/pmc_$c.csv; rm -rf gpurun_out/prof_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --steps 1 --warmup 1 --frames 30 --streams 1 --no-cpu-baseline --no-single --device-output > $P/pmc_chain_line_$c.json 2>/dev/null
  find gpurun_out/prof_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $P/pmc_chain_$c.csv; rm -rf gpurun_out/prof_$c
done
python3 tools/make_traffic_profile.py $P/pmc_FETCH_SIZE.csv $P/pmc_WRITE_SIZE.csv $P/r02_pmc_traffic.json $P/pmc_chain_FETCH_SIZE.csv $P/pmc_chain_WRITE_SIZE.csv $P/pmc_chain_line_FETCH_SIZE.json 2.0 > $P/pmc_summary.txt 2>&1
rm -f $P/pmc_*.csv
for f in $P/*.json; do echo $f; head -c 300 $f; echo; done
