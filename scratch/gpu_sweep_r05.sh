# round 5: random differential sweeps on the final tree (tools/gpu_sweep.py: generator configuration -> C ABI on the GPU -> bit-exact against the oracle)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sw5
date +%s > gpurun_out/sw5/t0
timeout 2400 python tools/gpu_sweep.py 400 91 > gpurun_out/sw5/sweep_a.log 2>&1; tail -2 gpurun_out/sw5/sweep_a.log; date +%s
timeout 2400 python tools/gpu_sweep.py 400 92 > gpurun_out/sw5/sweep_b.log 2>&1; tail -2 gpurun_out/sw5/sweep_b.log; date +%s
timeout 900 python tools/gpu_sweep.py 60 93 big > gpurun_out/sw5/sweep_big.log 2>&1; tail -2 gpurun_out/sw5/sweep_big.log; date +%s
