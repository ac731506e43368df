// jmcodec_amd/csrc/kernels.h -- host-callable launchers of the gfx950 kernels (kernels.hip, intra_lds.hip, deblock_lds.hip).
// Every launch processes a BATCH: d_pics / d_jobs are device arrays, blockIdx.y selects the picture; a picture takes part in a
// kernel only if the matching PS_* bit is set in PicParams::stages.
#pragma once
#include <hip/hip_runtime_api.h>
#include "jobs.h"

namespace jmamd {
void launch_recon_inter(const PicParams *d_pics, int n, int max_mbs, hipStream_t st);
void launch_recon_intra(const PicParams *d_pics, int n, hipStream_t st);          // spin-wait wavefront (sparse intra, any height)
void launch_deblock(const PicParams *d_pics, int n, hipStream_t st);              // spin-wait wavefront (any height)
bool intra_lds_supported(int mb_w, int mb_h);
void launch_intra_lds(const PicParams *d_pics, int n, int max_mb_h, int *progress, hipStream_t st);   // banded wavefront; progress: n * kDeblockProgressStride ints, cleared by this call
bool deblock_lds_supported(int mb_w, int mb_h);
constexpr int kDeblockMaxBands = 32, kDeblockProgressStride = 2 * kDeblockMaxBands;
// prep + LDS wavefront; progress: device array of n * kDeblockProgressStride ints (band step counters, cleared by this call)
void launch_deblock_lds(const PicParams *d_pics, int n, int max_mbs, int max_mb_h, int *progress, hipStream_t st);
// pitch-linear NV12 surface -> tight frame (out_fmt 0 = NV12, 1 = I420 order), nv_dec.cpp:782-820
void launch_packout(const PackJob *d_jobs, int n, int max_width, int max_height, hipStream_t st);
// tight I420 (fmt 1) / NV12 (fmt 0) frame in device memory -> ARGB32 in device memory
void launch_frame_to_argb(const uint8_t *d_src, int w, int h, int fmt, uint8_t *d_dst, int dst_pitch, hipStream_t st);
void launch_frame_to_nv12_pitch(const uint8_t *d_src, int w, int h, int fmt, uint8_t *d_dst, int pitch, hipStream_t st);   // tight I420 / NV12 -> pitch NV12 (encoder input)
}  // namespace jmamd
