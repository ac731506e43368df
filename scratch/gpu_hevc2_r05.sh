# round 5: HEVC intra block loop on scalar records -- parity tests, then 4K / 1080p lines head against the previous commit's library (scratch/_ab/prev)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; P=gpurun_out/hv2; mkdir -p $P
PREV=$GRAFT_REPO_ROOT/scratch/_ab/prev/libjm_amd_dec.so
timeout 1200 python -m pytest tests/test_hevc_gpu_parity.py -m gpu -x -q > $P/gputests.log 2>&1; tail -3 $P/gputests.log
for w in head prev; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $w = prev ] && L=$PREV
  JM_AMD_DEC_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h -- python3 bench.py --codec hevc --width 3840 --height 2160 --streams 16 --frames 16 --steps 2 --no-extra --no-cpu-baseline --no-single --device-output > $P/${w}_under_rocprof.json 2>/dev/null
  find gpurun_out/prof_h -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $P/${w}_hevc_4k_kernel_stats.csv; rm -rf gpurun_out/prof_h
  head -4 $P/${w}_hevc_4k_kernel_stats.csv | cut -d, -f1-7
done
