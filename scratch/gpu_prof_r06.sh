# scratch/gpu_prof_r06.sh -- round 6 profile set (run on the GPU box: gpurun -- 'bash scratch/gpu_prof_r06.sh [tag]'); results under gpurun_out/p3/
# Per BASELINE config: the plain bench line, the same run with device-resident output under rocprofv3 --kernel-trace --stats (the CSV must reproduce the
# line's roofline), and the two PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) with ONE stream -> profiles/r06_pmc_traffic_<config>.json.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r06}; P=gpurun_out/p6; mkdir -p $P
stats() {   # stats <name> <bench args...>: kernel stats CSV + the line of the profiled run
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k -- python3 bench.py "$@" --device-output --no-cpu-baseline --no-single --no-extra > $P/${T}_${n}_device_output_under_rocprof.json 2>/dev/null
  find gpurun_out/prof_k -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $P/${T}_${n}_kernel_stats.csv; rm -rf gpurun_out/prof_k
}
pmc() {     # pmc <tag> <bench args...>: one stream, one picture per launch (chain launches off)
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    JM_AMD_DEC_CHAIN_DEPTH=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_$c -- python3 bench.py "$@" --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-single --no-extra --device-output > /dev/null 2>&1
    find gpurun_out/prof_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $P/pmc_${tag}_$c.csv; rm -rf gpurun_out/prof_$c
  done
}
# ---- C1: H.264 Baseline 1080p (the driver's command) ----
timeout 900 python bench.py > $P/${T}_bench.json 2> $P/${T}_bench.err
stats c1
pmc c1 --frames 30
for c in FETCH_SIZE WRITE_SIZE; do    # chain launches on: k_chain
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --steps 1 --warmup 1 --frames 30 --streams 1 --no-cpu-baseline --no-single --no-extra --device-output > $P/pmc_c1chain_line_$c.json 2>/dev/null
  find gpurun_out/prof_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $P/pmc_c1chain_$c.csv; rm -rf gpurun_out/prof_$c
done
python3 tools/make_traffic_profile.py --fetch $P/pmc_c1_FETCH_SIZE.csv --write $P/pmc_c1_WRITE_SIZE.csv --out $P/${T}_pmc_traffic_h264_baseline_1920x1080.json --width 1920 --height 1080 \
  --env "JM_AMD_DEC_CHAIN_DEPTH=1" --command "python3 bench.py --frames 30 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-single --device-output" \
  --chain-fetch $P/pmc_c1chain_FETCH_SIZE.csv --chain-write $P/pmc_c1chain_WRITE_SIZE.csv --chain-line $P/pmc_c1chain_line_FETCH_SIZE.json > $P/pmc_summary_c1.txt 2>&1
# ---- C2: H.264 High 4K I B B P (CABAC, 8x8 transform) ----
C2="--tools high_b --width 3840 --height 2160"
timeout 600 python bench.py $C2 --streams 16 --frames 24 --steps 3 --no-cpu-baseline --no-single --no-extra > $P/${T}_c2_4k.json 2> $P/${T}_c2_4k.err
stats c2_4k $C2 --streams 16 --frames 24 --steps 3
pmc c2 $C2 --frames 24
python3 tools/make_traffic_profile.py --fetch $P/pmc_c2_FETCH_SIZE.csv --write $P/pmc_c2_WRITE_SIZE.csv --out $P/${T}_pmc_traffic_h264_high_b_3840x2160.json --width 3840 --height 2160 \
  --env "JM_AMD_DEC_CHAIN_DEPTH=1" --command "python3 bench.py $C2 --frames 24 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-single --device-output" > $P/pmc_summary_c2.txt 2>&1
# ---- C3: HEVC Main, 1080p and 4K ----
for sz in "1920 1080 32" "3840 2160 16"; do
  set -- $sz; w=$1; h=$2; f=$3
  C3="--codec hevc --width $w --height $h"
  timeout 600 python bench.py $C3 --streams 16 --frames $f --steps 3 --no-cpu-baseline --no-single --no-extra > $P/${T}_hevc_bench_${w}x${h}.json 2> $P/${T}_hevc_bench_${w}x${h}.err
  stats hevc_${w}x${h} $C3 --streams 16 --frames $f --steps 3
  pmc hevc_${w}x${h} $C3 --frames $f
  python3 tools/make_traffic_profile.py --fetch $P/pmc_hevc_${w}x${h}_FETCH_SIZE.csv --write $P/pmc_hevc_${w}x${h}_WRITE_SIZE.csv --out $P/${T}_pmc_traffic_hevc_${w}x${h}.json --width $w --height $h \
    --command "python3 bench.py $C3 --frames $f --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-single --device-output" > $P/pmc_summary_hevc_${w}x${h}.txt 2>&1
done
rm -f $P/pmc_*.csv
for f in $P/*.json; do echo $f; head -c 240 $f; echo; done
