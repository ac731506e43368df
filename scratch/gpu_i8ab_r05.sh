#!/bin/bash
# round 5: Intra8x8 in registers -- the all-intra probe (md5 of the output against the previous library's), then the H.264 GPU parity tests and a High sweep
cd "$GRAFT_REPO_ROOT" || exit 1
bash scratch/gpu_i8probe_r05.sh 2>&1 | grep -E "^==|k_intra_band" 
O=gpurun_out/i8; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/gputests.log 2>&1; tail -3 $O/gputests.log
timeout 1200 python tools/gpu_sweep.py 150 50505 > $O/sweep.log 2>&1; tail -3 $O/sweep.log
echo finished
