# round 3: the whole GPU suite, smoke(), the 2-ranks-on-one-GPU run, then the profile set of every BASELINE config (scratch/gpu_prof_r03.sh)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/p3
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/p3/r03_gputests.log 2>&1; tail -3 gpurun_out/p3/r03_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --streams 8 > gpurun_out/p3/r03_bench_2ranks_one_gpu.json 2>/dev/null
timeout 300 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/p3/r03_bench_s1.json 2>/dev/null
timeout 300 python bench.py --streams 8 --no-cpu-baseline > gpurun_out/p3/r03_bench_s8.json 2>/dev/null
timeout 300 python bench.py --tools high --no-cpu-baseline --no-single > gpurun_out/p3/r03_bench_high.json 2>/dev/null
timeout 300 python bench.py --tools high_b --no-cpu-baseline --no-single > gpurun_out/p3/r03_bench_high_b.json 2>/dev/null
timeout 300 python bench.py --codec hevc --streams 1 --frames 32 --steps 3 --no-cpu-baseline > gpurun_out/p3/r03_hevc_bench_s1.json 2>/dev/null
timeout 300 python bench.py --tools paff --no-cpu-baseline --no-single > gpurun_out/p3/r03_bench_paff.json 2>/dev/null
timeout 300 python bench.py --tools paff_b --no-cpu-baseline --no-single > gpurun_out/p3/r03_bench_paff_b.json 2>/dev/null
bash scratch/gpu_prof_r03.sh r03
