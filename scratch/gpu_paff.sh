# round 3: PAFF cases on the device (+ the lone-field case), then the chain tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/paff
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "paff or lone or field" > gpurun_out/paff/t1.log 2>&1; tail -15 gpurun_out/paff/t1.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain_launch" > gpurun_out/paff/t2.log 2>&1; tail -5 gpurun_out/paff/t2.log
