set -x
timeout 900 python -m pytest tests/test_hevc_gpu_parity.py -x -q 2>&1 | tail -15
