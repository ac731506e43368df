#!/bin/bash
# usage: tools/kernel_regs.sh file.hip ... : LDS bytes, VGPRs, spills and scratch per kernel (device-only compile for gfx950, in a temporary directory)
T=$(mktemp -d); trap 'rm -rf $T' EXIT
for src in "$@"; do
  b=$(basename $src .hip); d=$(cd $(dirname $src) && pwd)
  (cd $T && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c $d/$b.hip -o $b.o --save-temps=obj >/dev/null 2>&1)
  grep -E "^\s+\.(name|vgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):" $T/$b-hip-amdgcn-amd-amdhsa-gfx950.s | awk '{printf "%s ", $0} /\.vgpr_spill_count/ {print ""}' | sed 's/ \+/ /g'
done
