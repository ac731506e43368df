# round 4: large random sweeps on the FINAL tree (after the HEVC chroma pairing): 2 x 400 + 60 large configurations per codec
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sw3
timeout 2400 python tools/gpu_sweep.py 400 81 > gpurun_out/sw3/sweep_a.log 2>&1; tail -n 2 gpurun_out/sw3/sweep_a.log
timeout 2400 python tools/gpu_sweep.py 400 82 > gpurun_out/sw3/sweep_b.log 2>&1; tail -n 2 gpurun_out/sw3/sweep_b.log
timeout 1500 python tools/gpu_sweep.py 60 83 big > gpurun_out/sw3/sweep_big.log 2>&1; tail -n 2 gpurun_out/sw3/sweep_big.log
