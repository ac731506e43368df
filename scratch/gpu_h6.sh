cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hevc_gpu_parity.py -x -q 2>&1 | tail -2
mkdir -p gpurun_out/p
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --codec hevc --steps 1 --warmup 1 --frames 16 --streams 1 --width 1920 --height 1080 --no-cpu-baseline > /dev/null 2>&1
  find gpurun_out/prof_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/p/hevc_pmc_$c.csv; rm -rf gpurun_out/prof_$c
done
python3 tools/make_traffic_profile.py gpurun_out/p/hevc_pmc_FETCH_SIZE.csv gpurun_out/p/hevc_pmc_WRITE_SIZE.csv gpurun_out/p/r01_hevc_pmc_traffic.json | python3 -c "
import json,sys; d=json.load(sys.stdin); print({k:(v['fetch_raw'], v['write']) for k,v in d.items()})"
rm -f gpurun_out/p/hevc_pmc_*.csv
timeout 300 python bench.py --codec hevc --streams 16 --frames 32 --width 1920 --height 1080 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['host_ms_per_picture'], {k:(v['avg_us'], v['pictures_per_launch']) for k,v in d['kernels'].items()})"
