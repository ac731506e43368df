// jmcodec_amd/csrc/intra_lds.hip -- intra prediction + reconstruction (H.264 8.3), LDS-resident lockstep wavefront.
//
// Part of the replacement for cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:33-41).
//
// Intra prediction of MB(x,y) needs the UNFILTERED reconstructed samples of its left, top-left, top and
// top-right neighbours, so macroblocks with equal s = x + 2y are independent and steps run in order
// (the same diagonal as deblock_lds.hip).  Residuals do not depend on prediction: k_recon_inter computes them
// for intra macroblocks too (dequant + DC Hadamard + inverse transform) into a per-MB int16 buffer, so the
// serial part here is prediction + add only.
//
// grid = 2 workgroups x 512 threads: block 0 luma, block 1 chroma; 16 lanes per macroblock, 32 macroblock rows
// in flight per block, one raw s_barrier per step.  Neighbour samples never come from HBM: each macroblock
// row keeps in LDS the right column of its last macroblock and a 4-slot ring of bottom rows; macroblocks that
// are not intra (inter / I_PCM, already reconstructed by k_recon_inter) just refresh those from prefetched
// samples.  Intra4x4 is branch-free: every mode is "copy / 2-tap / 3-tap around centre c" on the 15-entry edge
// path  L3' L3 L2 L1 L0 TL T0..T7 T7'  and (c, kind) comes from a 9x16 table built in LDS at kernel start.
#include <hip/hip_runtime.h>
#include "jobs.h"
#include "kernels.h"
#include "kernel_common.h"
#include "chain_common.h"
#include "intra_device.h"     // tables, LDS layout, intra_luma_mb / intra_chroma_mb, intra_band_body (shared with chain.hip)

namespace jmamd {

// ------------------------------------------------------------------------------------------
// k_intra_band: the stage kernel -- grid (2 x bands, pictures); ctl = the batch's control buffer (chain_common.h layout), err = its error words
__global__ __launch_bounds__(kIBandRows * 16) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_intra_band(const PicParams *pics, int *ctl, int *err) {
    __shared__ __align__(16) uint8_t smem[kIntraSmemBytes];
    const PicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & PS_INTRA_LDS)) return;
    intra_band_body<false>(pp, blockIdx.x >> 1, blockIdx.x & 1, ctl + (size_t)blockIdx.y * kChainStride + kChainIntraRing, smem, nullptr, err + blockIdx.y);
}

bool intra_lds_supported(int mb_w, int mb_h) { return mb_w > 0 && mb_h <= kIBandRows * kDeblockMaxBands; }

// ctl must have been cleared for this batch (Engine::launch)
void launch_intra_lds(const PicParams *d_pics, int n, int max_mb_h, int *ctl, int *err, hipStream_t st) {
    const int bands = (max_mb_h + kIBandRows - 1) / kIBandRows;
    hipLaunchKernelGGL(k_intra_band, dim3(2 * bands, n), dim3(kIBandRows * 16), 0, st, d_pics, ctl, err);
}

}  // namespace jmamd
