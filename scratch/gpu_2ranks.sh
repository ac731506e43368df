# the driver's multi-rank command with 2 ranks on the one GPU of the box: the ranks must see each other (KFD sysfs) and form no chain launches
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --streams 8 > gpurun_out/r03_bench_2ranks_one_gpu.json 2> gpurun_out/r03_2ranks.err
grep -v "amdgpu.ids\|Warning\|warn" gpurun_out/r03_2ranks.err | tail -8
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r03_bench_2ranks_one_gpu.json").read().strip().splitlines()[-1])
print("value", l["value"], "n_gpus", l["n_gpus"], "bit_exact", l["bit_exact"], "errors", l["decode_errors"], l["engine"]["chain_recoveries_whole_run"], l["engine"]["gpu_shared_with_another_process"], l["engine"]["chain_batches_whole_run"])
PY
python bench.py --streams 8 --steps 4 --no-cpu-baseline --no-single 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('alone: value', l['value'], l['engine']['chain_recoveries_whole_run'], l['engine']['gpu_shared_with_another_process'], l['engine']['chain_batches_whole_run'])"
