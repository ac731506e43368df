# round 3: PAFF on the device: parity cases, full size, lone field, mixed batches, then a random sweep with field pictures
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/paff
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "paff or partner or field" > gpurun_out/paff/t1.log 2>&1; tail -15 gpurun_out/paff/t1.log
timeout 1200 python tools/gpu_sweep.py ${1:-60} 31 > gpurun_out/paff/sweep.log 2>&1; tail -6 gpurun_out/paff/sweep.log
