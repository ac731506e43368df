# round 3, first GPU call: the GPU suite on the tree as it stands, smoke(), then the profile set
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputests_first.log 2>&1; tail -3 gpurun_out/r03_gputests_first.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash scratch/gpu_prof_r03.sh r03a
