# round 4: random sweeps with fresh seeds on the later kernel set (one window per macroblock, segment stores, napping waits)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sw2
timeout 1500 python tools/gpu_sweep.py 200 51 > gpurun_out/sw2/sweep_a.log 2>&1; tail -2 gpurun_out/sw2/sweep_a.log
timeout 1500 python tools/gpu_sweep.py 200 52 > gpurun_out/sw2/sweep_b.log 2>&1; tail -2 gpurun_out/sw2/sweep_b.log
timeout 900 python tools/gpu_sweep.py 30 53 big > gpurun_out/sw2/sweep_big.log 2>&1; tail -2 gpurun_out/sw2/sweep_big.log
timeout 600 python -m pytest tests -m gpu -x -q -rs -k "diagnostic or chain" 2>&1 | tail -3
