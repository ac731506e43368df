#!/bin/bash
# Developer script (gpurun): the host parser alone on the GPU box's CPU -- variants of tools/_build/host_bench* built beforehand, one thread, parse only.
mkdir -p gpurun_out
python - <<'PY'
import sys; sys.path.insert(0, '.')
from tools import streams
open('/tmp/c3.hevc', 'wb').write(streams.generate_hevc(**streams.config_c3(frames=32, width=1920, height=1080, stream_id=0)))
cfg = streams.config_c1(stream_id=0, frames=60, width=1920, height=1080); cfg.update(cabac=1, t8x8=1)
open('/tmp/high.h264', 'wb').write(streams.generate(**cfg))
cfg.update(bframes=2, num_ref=2, poc_type=0)
open('/tmp/high_b.h264', 'wb').write(streams.generate(**cfg))
PY
{
grep -m1 "model name" /proc/cpuinfo; nproc
for b in tools/_build/host_bench*; do
  for i in 1 2 3; do echo "$b hevc: $(JM_AMD_DEC_THREADS=1 $b /tmp/c3.hevc 5 1 | tail -1)"; done
  for f in high high_b; do for i in 1 2; do echo "$b $f: $(JM_AMD_DEC_THREADS=1 $b /tmp/$f.h264 3 0 | tail -1)"; done; done
done
} > gpurun_out/hostbench.txt 2>&1
cat gpurun_out/hostbench.txt
