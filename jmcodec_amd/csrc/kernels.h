// jmcodec_amd/csrc/kernels.h -- host-callable launchers of the gfx950 kernels (kernels.hip, intra_lds.hip, deblock_lds.hip).
// Every launch processes a BATCH: d_pics / d_jobs are device arrays, blockIdx.y selects the picture; a picture takes part in a
// kernel only if the matching PS_* bit is set in PicParams::stages.
#pragma once
#include <hip/hip_runtime_api.h>
#include "jobs.h"

namespace jmamd {
void launch_recon_inter(const PicParams *d_pics, int n, int max_mbs, bool any_bipred, bool any_field, int *d_err, hipStream_t st);
void launch_recon_intra(const PicParams *d_pics, int n, hipStream_t st);          // spin-wait wavefront (sparse intra, any height)
void launch_deblock(const PicParams *d_pics, int n, hipStream_t st);              // spin-wait wavefront (any height)
bool intra_lds_supported(int mb_w, int mb_h);
// ctl: the batch's control buffer, n * kChainStride ints (chain_common.h), cleared once per batch by the caller; err: the batch's error words
// (one int per picture, host-pinned and device-visible): a kernel whose bounded wait gives up writes a non-zero code there
void launch_intra_lds(const PicParams *d_pics, int n, int max_mb_h, int *ctl, int *err, hipStream_t st);   // banded wavefront
bool deblock_lds_supported(int mb_w, int mb_h);
constexpr int kDeblockMaxBands = 32, kDeblockProgressStride = 2 * kDeblockMaxBands;
// boundary strengths -> DbRec (PS_DEBLOCK_LDS or PS_CHAIN pictures)
void launch_deblock_prep(const PicParams *d_pics, int n, int max_mbs, hipStream_t st);
// debug_stall: test hook -- the bands never publish their step counters, so every band below the first one runs into its bounded wait
void launch_deblock_lds(const PicParams *d_pics, int n, int max_mb_h, int *ctl, int *err, bool debug_stall, hipStream_t st);   // LDS wavefront (after the prep)
// chain launch (chain.hip): reconstruction + deblocking of every PS_CHAIN picture of the batch in ONE kernel; pictures of a stream follow each
// other at macroblock granularity (chain_common.h).  Pictures must be ordered so that a picture's in-launch references have a lower index.
bool chain_supported(int mb_w, int mb_h);
// d_groups: the work list, n_groups entries `picture << 16 | kind << 15 | index` (kind 0: reconstruction of 8 macroblocks, index = row * 32 +
// segment; kind 1: deblocking band `index`), in the order in which the dispatcher shall start them (chain.hip)
void launch_chain(const PicParams *d_pics, const uint32_t *d_groups, int n_groups, bool with_intra, int *ctl, int *err, bool debug_stall, hipStream_t st);
int  chain_band_rows();
int  deblock_row_lag();        // steps between macroblock rows of the deblocking wavefront (chain_common.h kRowLag)
// workgroups of k_chain / k_chain_i the CURRENT device keeps resident (0: unknown); the engine bounds a launch's bands by half of it
int  chain_resident_workgroups(bool intra);
int  chain_ctl_ints();                                                                         // kChainStride
int  chain_tail_head_ints(); int chain_tail_wait_limit();                                           // kChainTailHead, kTailWaitLimit (chain_common.h)
int  chain_tail_ints();                                                                        // kChainTail: ints behind the pictures' blocks
// pitch-linear NV12 surface -> tight frame (out_fmt 0 = NV12, 1 = I420 order), nv_dec.cpp:782-820
void launch_packout(const PackJob *d_jobs, int n, int max_width, int max_height, hipStream_t st);
// tight I420 (fmt 1) / NV12 (fmt 0) frame in device memory -> ARGB32 in device memory
void launch_frame_to_argb(const uint8_t *d_src, int w, int h, int fmt, uint8_t *d_dst, int dst_pitch, hipStream_t st);
// tight I420 / NV12 -> pitch NV12 (encoder input)
void launch_frame_to_nv12_pitch(const uint8_t *d_src, int w, int h, int fmt, uint8_t *d_dst, int pitch, hipStream_t st);
}  // namespace jmamd
