// jmcodec_amd/csrc/kernels.h -- host-callable launchers of the gfx950 kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include "jobs.h"

namespace jmamd {
void launch_recon_inter(const PicParams &pp, hipStream_t st);
void launch_recon_intra(const PicParams &pp, hipStream_t st);
void launch_deblock(const PicParams &pp, hipStream_t st);            // spin-wait wavefront (any picture height)
// LDS-resident lockstep intra wavefront (intra_lds.hip); resid: 768 B per macroblock written by k_recon_inter
bool intra_lds_supported(int mb_w, int mb_h);
void launch_intra_lds(const PicParams &pp, const void *resid, hipStream_t st);
// LDS-resident lockstep wavefront (deblock_lds.hip); dbrec_scratch: device buffer of 96 B per macroblock
bool deblock_lds_supported(int mb_w, int mb_h);
void launch_deblock_lds(const PicParams &pp, void *dbrec_scratch, hipStream_t st);
// src: pitch-linear NV12 surface; dst: tight frame (out_fmt 0 = NV12, 1 = I420 order) of width x height
void launch_packout(const uint8_t *src, int pitch, int chroma_offset, int width, int height, int out_fmt,
                    uint8_t *dst, hipStream_t st);
}  // namespace jmamd
