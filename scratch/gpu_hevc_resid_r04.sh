# round 4: k_hevc_resid takes the Cb and Cr blocks of a transform unit in one workgroup (dword stores): HEVC tests, a sweep, kernel time before / after is read
# from the C3 1080p kernel-stats run (profiles/r04_hevc_1920x1080_kernel_stats.csv holds the "before")
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; P=gpurun_out/hr; mkdir -p $P
timeout 900 python -m pytest tests -m gpu -x -q -rs -k "hevc or Hevc or HEVC" 2>&1 | tail -3
timeout 1200 python tools/gpu_sweep.py 120 71 > $P/sweep.log 2>&1; tail -n 2 $P/sweep.log
for w in 1920x1080:32 3840x2160:16; do
  s=${w%%:*}; f=${w##*:}; W=${s%%x*}; H=${s##*x}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k -- python3 bench.py --codec hevc --width $W --height $H --streams 16 --frames $f --steps 3 --device-output --no-cpu-baseline --no-single --no-extra > $P/hevc_${s}.json 2>/dev/null
  find gpurun_out/prof_k -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $P/hevc_${s}_kernel_stats.csv; rm -rf gpurun_out/prof_k
  head -8 $P/hevc_${s}_kernel_stats.csv | cut -d, -f1-5 | cut -c1-120
done
