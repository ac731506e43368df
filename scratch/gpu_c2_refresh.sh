# C2 (H.264 High 4K I B B P, 16 streams) again after the B-picture parser change: the plain line and the rocprofv3 kernel-stats pair
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=gpurun_out/p3; mkdir -p $P; T=r03
C2="--tools high_b --width 3840 --height 2160"
timeout 600 python bench.py $C2 --streams 16 --frames 24 --steps 3 --no-cpu-baseline --no-single > $P/${T}_c2_4k.json 2> $P/${T}_c2_4k.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k -- python3 bench.py $C2 --streams 16 --frames 24 --steps 3 --device-output --no-cpu-baseline --no-single > $P/${T}_c2_4k_device_output_under_rocprof.json 2>/dev/null
find gpurun_out/prof_k -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $P/${T}_c2_4k_kernel_stats.csv; rm -rf gpurun_out/prof_k
python - <<'PY'
import json
for f in ('gpurun_out/p3/r03_c2_4k.json', 'gpurun_out/p3/r03_c2_4k_device_output_under_rocprof.json'):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['scaling_bound'], d['host_cpu']['cpu_ms_per_frame'], d['host_cpu']['cpus_busy'], d['bit_exact'])
PY
head -8 $P/${T}_c2_4k_kernel_stats.csv | cut -c1-150
