cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2zs
timeout 900 python -m pytest tests/test_hevc_gpu_parity.py -m gpu -x -q > gpurun_out/r2zs/pytest.txt 2>&1; tail -3 gpurun_out/r2zs/pytest.txt
timeout 300 python bench.py --codec hevc --no-cpu-baseline --no-single --steps 3 > gpurun_out/r2zs/hevc.json 2>/dev/null || echo FAIL
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r2zs/hevc.json").read().strip().splitlines()[-1])
print(d["value"], d["bit_exact"], d["host_cpu"]["cpus_busy"], d["kernels"])
PY
for s in 1 4 8; do timeout 300 python bench.py --codec hevc --streams $s --no-cpu-baseline --no-single --steps 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s$s', d['value'], d['bit_exact'], d['kernels']['k_intra'], d['kernels']['k_inter'])"; done
