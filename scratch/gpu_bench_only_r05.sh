#!/bin/bash
# round 5: the driver's command alone (after the PMC traffic files of the final set are in the tree: the line reads them)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/b5
timeout 1200 python bench.py > gpurun_out/b5/r05_bench.json 2> gpurun_out/b5/r05_bench.err; tail -c 600 gpurun_out/b5/r05_bench.err; head -c 400 gpurun_out/b5/r05_bench.json; echo; echo finished
