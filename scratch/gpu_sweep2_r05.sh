#!/bin/bash
# round 5: two more random sweeps on the final tree (fresh seeds)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/sw2
timeout 2400 python tools/gpu_sweep.py 400 94 > gpurun_out/sw2/sweep_c.log 2>&1; tail -2 gpurun_out/sw2/sweep_c.log
timeout 2400 python tools/gpu_sweep.py 400 95 > gpurun_out/sw2/sweep_d.log 2>&1; tail -2 gpurun_out/sw2/sweep_d.log
echo finished
