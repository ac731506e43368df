cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
JM_AMD_DEC_EXP_NOPACK=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -- python3 bench.py --no-cpu-baseline --steps 3 > /dev/null 2>&1
f=$(find gpurun_out/prof_t -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py $f 2>&1 | head -40
rm -rf gpurun_out/prof_t
