#!/bin/bash
# Developer script (gpurun): sampling profile of the host parser (tools/host_bench, JM_HOST_BENCH_PROF) on the GPU box's CPU; symbolised on the box
mkdir -p gpurun_out
python - <<'PY'
import sys; sys.path.insert(0, '.')
from tools import streams
open('/tmp/c3.hevc', 'wb').write(streams.generate_hevc(**streams.config_c3(frames=32, width=1920, height=1080, stream_id=0)))
cfg = streams.config_c1(stream_id=0, frames=60, width=1920, height=1080); cfg.update(cabac=1, t8x8=1, bframes=2, num_ref=2, poc_type=0)
open('/tmp/high_b.h264', 'wb').write(streams.generate(**cfg))
PY
cd /tmp
JM_AMD_DEC_THREADS=1 JM_HOST_BENCH_PROF=/tmp/prof_hevc.txt $GRAFT_REPO_ROOT/tools/_build/host_bench /tmp/c3.hevc 8 1
JM_AMD_DEC_THREADS=1 JM_HOST_BENCH_PROF=/tmp/prof_h264.txt $GRAFT_REPO_ROOT/tools/_build/host_bench /tmp/high_b.h264 8 0
cd $GRAFT_REPO_ROOT
python scratch/hostprof.py /tmp/prof_hevc.txt tools/_build/host_bench > gpurun_out/hostprof_hevc.txt 2>&1
python scratch/hostprof.py /tmp/prof_h264.txt tools/_build/host_bench > gpurun_out/hostprof_h264.txt 2>&1
for f in "decode_sub_block" "HevcPicParser::residual_coding(" "HevcPicParser::coding_unit(" "HevcPicParser::prediction_unit(" "HevcPicParser::finish_picture()"; do
  python scratch/hostprof_lines.py /tmp/prof_hevc.txt "$f" | sort -rn | head -25 >> gpurun_out/hostprof_hevc_lines.txt; done
head -40 gpurun_out/hostprof_hevc.txt
