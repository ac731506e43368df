#!/bin/bash
# round 6: per-function shares of the host half (one parse worker, development container) for C1 / C2 / C3
set -e
make -C tools host_bench > /dev/null
T=/tmp/hostprof_r06; mkdir -p $T
python - <<'PY'
import sys; sys.path.insert(0, '.')
from tools import streams
T='/tmp/hostprof_r06'
open(T+'/c1.h264','wb').write(streams.generate(**streams.config_c1(frames=60)))
c2=streams.config_c1(stream_id=0, frames=24, width=3840, height=2160); c2.update(cabac=1, t8x8=1, bframes=2, num_ref=2, poc_type=0)
open(T+'/c2.h264','wb').write(streams.generate(**c2))
open(T+'/c3.h265','wb').write(streams.generate_hevc(**streams.config_c3(frames=16)))
PY
for c in "c1 h264 0 4" "c2 h264 0 2" "c3 h265 1 2"; do set -- $c
  JM_AMD_DEC_THREADS=1 JM_HOST_BENCH_PROF=$T/$1.samples tools/_build/host_bench $T/$1.$2 $4 $3 > $T/$1.rate
  { echo "# round 6, scratch/host_profiles_r06.sh: host half of $1 (one parse worker, JM_AMD_DEC_THREADS=1, development container: Xeon 2.6 GHz; the GPU boxes' EPYC 9575F is ~1.9 x faster), tools/host_bench + SIGPROF every 50 us, symbolised by scratch/hostprof.py"; cat $T/$1.rate; python scratch/hostprof.py $T/$1.samples; } > profiles/r06_host_profile_$1.txt
done
head -30 profiles/r06_host_profile_c3.txt
