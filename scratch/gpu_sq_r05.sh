# round 5: SQ counters per kernel and picture for C1 / C2 / C3 (one stream, chain launches off) -> gpurun_out/sq5/<tag>_<config>.json
# usage: bash scratch/gpu_sq_r05.sh <tag> <library path relative to the repo root> [configs...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; TAG=$1; LIB=$GRAFT_REPO_ROOT/$2; shift; shift; CFGS="${@:-c1 c2 c3}"
P=gpurun_out/sq5; mkdir -p $P
for cfg in $CFGS; do
  case $cfg in
    c1) ARGS="--frames 30 --streams 1"; NAME="C1: H.264 Baseline 1080p I/P (CAVLC), one stream";;
    c2) ARGS="--tools high_b --width 3840 --height 2160 --frames 12 --streams 1"; NAME="C2: H.264 High 4K I B B P (CABAC, 8x8), one stream";;
    c3) ARGS="--codec hevc --width 3840 --height 2160 --frames 16 --streams 1"; NAME="C3: HEVC Main 4K (64x64 CTB, SAO + deblocking), one stream";;
  esac
  CMD="python3 bench.py --steps 1 --warmup 1 $ARGS --no-extra --no-cpu-baseline --no-single --device-output"
  FILES=""
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVES"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-30)
    JM_AMD_DEC_LIB=$LIB JM_AMD_DEC_CHAIN_DEPTH=1 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_sq -- $CMD > /dev/null 2> $P/err_${TAG}_${cfg}.txt
    f=$(find gpurun_out/prof_sq -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then cp $f $P/raw_${TAG}_${cfg}_$n.csv; FILES="$FILES $P/raw_${TAG}_${cfg}_$n.csv"; fi
    rm -rf gpurun_out/prof_sq
  done
  python3 tools/sq_counters_summary.py --out $P/${TAG}_${cfg}.json --config "$NAME" --command "JM_AMD_DEC_CHAIN_DEPTH=1 rocprofv3 --kernel-trace --pmc <group> -- $CMD" $FILES | tee $P/${TAG}_${cfg}.txt
  rm -f $P/raw_${TAG}_${cfg}_*.csv
done
