cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/p/gpu_tests_tail.txt
# H.264 headline: bench line, rocprof kernel stats, PMC traffic
timeout 400 python bench.py > gpurun_out/p/r01_bench.json 2> gpurun_out/p/r01_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_a -- python3 bench.py --no-cpu-baseline > gpurun_out/p/r01_bench_under_rocprof.json 2>/dev/null
find gpurun_out/prof_a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/p/r01_kernel_stats.csv; rm -rf gpurun_out/prof_a
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --steps 1 --warmup 1 --frames 30 --streams 1 --no-cpu-baseline > /dev/null 2>&1
  find gpurun_out/prof_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/p/pmc_$c.csv; rm -rf gpurun_out/prof_$c
done
python3 tools/make_traffic_profile.py gpurun_out/p/pmc_FETCH_SIZE.csv gpurun_out/p/pmc_WRITE_SIZE.csv gpurun_out/p/r01_pmc_traffic.json > /dev/null
# HEVC (config C3): bench lines and rocprof kernel stats
timeout 400 python bench.py --codec hevc --streams 16 --frames 32 --width 1920 --height 1080 --steps 3 --warmup 1 > gpurun_out/p/r01_hevc_bench_1080p.json 2>/dev/null
timeout 400 python bench.py --codec hevc --streams 16 --frames 16 --width 3840 --height 2160 --steps 3 --warmup 1 > gpurun_out/p/r01_hevc_bench_4k.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h -- python3 bench.py --codec hevc --streams 16 --frames 16 --width 3840 --height 2160 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/p/r01_hevc_bench_4k_under_rocprof.json 2>/dev/null
find gpurun_out/prof_h -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/p/r01_hevc_kernel_stats.csv; rm -rf gpurun_out/prof_h
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --codec hevc --steps 1 --warmup 1 --frames 16 --streams 1 --width 1920 --height 1080 --no-cpu-baseline > /dev/null 2>&1
  find gpurun_out/prof_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/p/hevc_pmc_$c.csv; rm -rf gpurun_out/prof_$c
done
python3 tools/make_traffic_profile.py gpurun_out/p/hevc_pmc_FETCH_SIZE.csv gpurun_out/p/hevc_pmc_WRITE_SIZE.csv gpurun_out/p/r01_hevc_pmc_traffic.json > /dev/null
rm -f gpurun_out/p/pmc_*.csv gpurun_out/p/hevc_pmc_*.csv
for f in gpurun_out/p/*.json; do echo $f; head -c 400 $f; echo; done
for t in high high_b; do timeout 300 python bench.py --tools $t --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tools $t', d['value'], d['host_ms_per_picture'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items()})"; done
