#!/bin/bash
# usage: regs.sh file.hip ... : VGPRs / spills / scratch per kernel (device-only compile for gfx950)
for src in "$@"; do
  b=$(basename $src .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c $src -o $b.o --save-temps=obj >/dev/null 2>&1
  grep -E "^\s+\.(name|vgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|lds_size|group_segment_fixed_size):" $b-hip-amdgcn-amd-amdhsa-gfx950.s | awk '{printf "%s ", $0} /\.vgpr_spill_count/ {print ""}' | sed 's/ \+/ /g'
done
