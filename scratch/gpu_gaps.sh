# round 3: gaps in frame_num + PAFF on the device: parity cases, then a sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/gaps
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gaps or paff or partner or field" > gpurun_out/gaps/t1.log 2>&1; tail -5 gpurun_out/gaps/t1.log
timeout 1500 python tools/gpu_sweep.py ${1:-100} 41 > gpurun_out/gaps/sweep.log 2>&1; tail -3 gpurun_out/gaps/sweep.log
