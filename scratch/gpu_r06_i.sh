#!/bin/bash
# round 6: what takes a chain launch's waves off the device for ~25 ms at the end of a pass?  16 streams x 20 steps, six runs each:
#   fresh   the direct output route, a fresh output buffer per pass freed when the pass ends (rounds 1-5)
#   reused  the direct output route, one buffer per handle for the whole run
#   pinned  JM_AMD_DEC_OUT_PINNED=1 (no page-locking of caller memory), fresh buffers
#   devout  --device-output (no output copies at all)
mkdir -p gpurun_out/i; P=gpurun_out/i
B="--no-extra --no-cpu-baseline --no-single --steps 20 --streams 16"
for r in 1 2 3 4 5 6; do
  JM_BENCH_FRESH_BUFFERS=1 JM_AMD_DEC_VERBOSE=1 timeout 600 python bench.py $B > $P/fresh_$r.json 2> $P/fresh_$r.err
  JM_AMD_DEC_VERBOSE=1 timeout 600 python bench.py $B > $P/reused_$r.json 2> $P/reused_$r.err
  JM_BENCH_FRESH_BUFFERS=1 JM_AMD_DEC_OUT_PINNED=1 JM_AMD_DEC_VERBOSE=1 timeout 600 python bench.py $B > $P/pinned_$r.json 2> $P/pinned_$r.err
  JM_AMD_DEC_VERBOSE=1 timeout 600 python bench.py $B --device-output > $P/devout_$r.json 2> $P/devout_$r.err
done
for k in fresh reused pinned devout; do echo "$k: runs with clock gaps $(grep -l 'clock gap' $P/${k}_*.err | wc -l) of $(ls $P/${k}_*.err | wc -l); gaps $(grep -h 'clock gap' $P/${k}_*.err | wc -l)"; done
grep -h "clock gap" $P/*.err | cut -c1-230 | head -30
mkdir -p gpurun_out/h3; bash scratch/gpu_soak_r06.sh > gpurun_out/h3/soak.txt 2>&1
tail -8 gpurun_out/h3/soak.txt | cut -c1-300
