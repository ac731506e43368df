// jmcodec_amd/csrc/chain_intra.hip -- k_chain_i: the chain kernel (chain.hip) with a third role, the intra wavefront, so that the I picture of an IDR period
// runs as the FIRST picture of a chain launch instead of alone on the intra lane: its reconstruction groups (or k_recon_inter before the launch) compute the
// residuals, its intra bands (intra_device.h) predict and reconstruct, its deblocking bands follow the intra wavefront four steps behind -- intra prediction
// reads UNFILTERED neighbours, so macroblock (x, y) may be deblocked once the intra wavefront has passed (x + 1, y + 1) -- and the P pictures behind it follow
// its deblocking like that of any other picture.  Part of the replacement for cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:33-41).
//
// A translation unit of its own, compiled for whatever occupancy the three roles together allow (the register allocation of a fused kernel is the maximum
// over its roles plus what lives across them: 179 VGPRs, 2 waves per SIMD); Engine::form bounds the bands of such a launch accordingly.  P-only launches keep
// using k_chain (167 VGPRs, 3 waves per SIMD).
#include <hip/hip_runtime.h>
#include "jobs.h"
#include "kernels.h"
#include "kernel_common.h"
#include "chain_common.h"
#include "recon_device.h"
#include "deblock_device.h"
#include "intra_device.h"

namespace jmamd {

int deblock_depth();

constexpr int kChainISmemBytes = kIntraSmemBytes > kDeblockSmemBytes ? (kIntraSmemBytes > (int)sizeof(ReconLds) ? kIntraSmemBytes : (int)sizeof(ReconLds))
                                                                      : (kDeblockSmemBytes > (int)sizeof(ReconLds) ? kDeblockSmemBytes : (int)sizeof(ReconLds));

template <int DEPTH>
__global__ __launch_bounds__(256, 3) __attribute__((flatten)) void k_chain_i(const PicParams *pics, int *ctl, int *err, const uint32_t *groups, int pub) {
    // one LDS block for whichever role this workgroup has (separate static arrays would add up to 65 KB)
    __shared__ __align__(16) uint8_t smem[kChainISmemBytes];
    const int g = (int)blockIdx.x >> 1, rem = (int)blockIdx.x & 1;
    const uint32_t entry = groups[g];
    const PicParams &pp = pics[entry >> 16];
    const bool census_on = pub > 0 && (pub & 0x10000) != 0;
    if (pub > 0) pub &= 0xffff;
    ChainView cv{ctl, err}; cv.census_on = census_on;
    cv.census(ChainView::CENSUS_MAX_GROUP, g);
    if (!(entry & 0x8000u)) {
        const int row = (int)(entry & 0x7fffu) >> 5, seg = (int)entry & 31;
        const int x = seg * 8 + rem * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform, and known to be (recon_device.h)
        const bool valid = row < pp.mb_h && x < pp.mb_w;
        if (row >= pp.mb_h || seg * 8 + rem * 4 >= pp.mb_w) return;
        cv.census(ChainView::CENSUS_RECON_STARTED); cv.stamp(pp.chain_idx, ChainView::STAMP_RECON_FIRST);
        const uint32_t tr0 = census_on ? (uint32_t)wall_clock64() : 0u;
        recon_inter_wave<true, true, true, false>(pp, valid ? row * pp.mb_w + x : 0, valid, *reinterpret_cast<ReconLds *>(smem), cv);
        cv.census(ChainView::CENSUS_RECON_DONE); cv.stamp(pp.chain_idx, ChainView::STAMP_RECON_LAST);
        if (census_on) cv.census(ChainView::CENSUS_RECON_TICKS, (int)((uint32_t)wall_clock64() - tr0));
    } else {
        cv.census(ChainView::CENSUS_BAND_STARTED); cv.stamp(pp.chain_idx, ChainView::STAMP_BAND_FIRST);
        int *cpic = cv.pic(pp.chain_idx);
        const int band = (int)(entry & 31u);
        if (entry & 0x4000u) intra_band_body<true>(pp, band, rem == 1, cpic + kChainIntraRing, smem, cpic, err + pp.chain_idx);
        else if (pp.stages & PS_CHAIN_INTRA) deblock_band_body<DEPTH, true, true>(pp, band, rem == 1, cpic + kChainRing, pub, smem, cpic, err + pp.chain_idx);
        else deblock_band_body<DEPTH, true, false>(pp, band, rem == 1, cpic + kChainRing, pub, smem, cpic, err + pp.chain_idx);
        cv.census(ChainView::CENSUS_BAND_DONE); cv.stamp(pp.chain_idx, ChainView::STAMP_BAND_LAST);
    }
}

// workgroups of k_chain_i the current device holds at once (occupancy x compute units, as the runtime reports them for THIS device: a compute
// partition or a smaller part holds fewer than a whole MI355X); 0 when the query fails
int chain_intra_resident_workgroups() {
    int per_cu = 0, dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    const bool d2 = deblock_depth() <= 2;
    if ((d2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chain_i<2>, 256, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chain_i<3>,
        256, 0)) != hipSuccess) return 0;
    return per_cu * prop.multiProcessorCount;
}

void launch_chain_intra(const PicParams *d_pics, const uint32_t *d_groups, int n_groups, int *ctl, int *err, int depth, int pub, hipStream_t st) {
    dim3 grid((unsigned)n_groups * 2u), block(256);
    if (depth <= 2) hipLaunchKernelGGL((k_chain_i<2>), grid, block, 0, st, d_pics, ctl, err, d_groups, pub);
    else hipLaunchKernelGGL((k_chain_i<3>), grid, block, 0, st, d_pics, ctl, err, d_groups, pub);
}

}  // namespace jmamd
