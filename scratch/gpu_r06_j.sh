#!/bin/bash
# round 6: do the clock gaps of the chain launches coincide with changes of the device's clock levels?  (read-only sysfs polling beside the bench)
mkdir -p gpurun_out/j; O=gpurun_out/j
ls /sys/class/drm/card*/device/pp_dpm_* 2>&1 | head -12 > $O/sysfs.txt
for r in 1 2 3 4 5 6; do
  python scratch/dpm_watch.py $O/dpm_$r.txt 40 &
  W=$!
  JM_BENCH_FRESH_BUFFERS=1 JM_AMD_DEC_VERBOSE=1 timeout 600 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --streams 3 > $O/run_$r.json 2> $O/run_$r.err
  JM_BENCH_FRESH_BUFFERS=1 JM_AMD_DEC_VERBOSE=1 timeout 600 python bench.py --no-extra --no-cpu-baseline --no-single --steps 20 --streams 16 > $O/run16_$r.json 2> $O/run16_$r.err
  kill $W 2>/dev/null; wait $W 2>/dev/null
done
cat $O/sysfs.txt
grep -h "clock gap" $O/*.err | cut -c150-400
for r in 1 2 3 4 5 6; do echo "== dpm_$r"; head -40 $O/dpm_$r.txt | cut -c1-200; done
