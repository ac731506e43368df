#!/bin/bash
# round 5: is the Intra8x8 path what makes a High I picture's k_intra_band step slower than a Baseline one?  All-intra streams, transform_8x8_mode_flag off / on,
# CAVLC / CABAC, 1080p and 4K, one stream through the native harness under rocprofv3 --kernel-trace --stats.  JM_AMD_DEC_LIB selects the library (A/B).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/i8; mkdir -p $O
python - <<'PY'
import sys
sys.path.insert(0, 'tools')
import streams
for name, w, h in (("1080", 1920, 1080), ("4k", 3840, 2160)):
    for t8 in (0, 1):
        open(f'/tmp/i_{name}_t{t8}.h264', 'wb').write(streams.generate(width=w, height=h, frames=6, qp=30, gop=1, seed=0x4A4D0555, mode=0, deblock=1, num_ref=1,
            level_idc=52, cabac=1, t8x8=t8, poc_type=0))
PY
make -C tools harness >/dev/null 2>&1
for lib in ${LIBS:-head}; do
  L=$GRAFT_REPO_ROOT/jmcodec_amd/lib/libjm_amd_dec.so; [ $lib != head ] && L=$GRAFT_REPO_ROOT/scratch/_ab/$lib/libjm_amd_dec.so
  for s in 1080_t0 1080_t1 4k_t0 4k_t1; do
    JM_AMD_DEC_LIB=$L LD_LIBRARY_PATH=$(dirname $L):$LD_LIBRARY_PATH timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_i8 -- tools/_build/test_amd_dec /tmp/i_$s.h264 /tmp/o_${lib}_$s.yuv --loops 3 > $O/${lib}_$s.log 2>&1
    f=$(find gpurun_out/prof_i8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${lib}_${s}_kernel_stats.csv; rm -rf gpurun_out/prof_i8
    echo "== $lib $s $(md5sum < /tmp/o_${lib}_$s.yuv | cut -c1-12)"; grep -E "k_intra_band|k_recon_inter|k_deblock_band" $O/${lib}_${s}_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-200
  done
done 2>&1 | tee $O/summary.txt
echo finished
