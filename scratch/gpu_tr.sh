JM_AMD_DEC_TRACE=gpurun_out/tr timeout 300 python bench.py --no-cpu-baseline --steps 3 > /dev/null 2>&1
python3 tools/host_trace.py gpurun_out/tr | sed -n 1,70p
rm -f gpurun_out/tr*.csv
