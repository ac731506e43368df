// jmcodec_amd/csrc/chain.hip -- k_chain: motion compensation + residual AND in-loop deblocking of a batch of P / B pictures in ONE launch,
// with consecutive pictures of the same stream pipelined at macroblock granularity (protocol: chain_common.h).
//
// Part of the replacement for cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:33-41): the reference decodes one picture at a time on
// a fixed-function pipeline; here a single stream keeps several pictures in flight on the CUs, because picture n+1's macroblock (x, y) only
// needs picture n deblocked a few macroblocks beyond (x, y), not the whole picture.
//
// grid = 1-D, cut into GROUPS of 2 workgroups; groups[g] = picture << 16 | kind << 15 | index says what group g does:
//   kind 0  reconstruction of 8 consecutive macroblocks of a row (index = row * 32 + segment): 4 macroblocks per workgroup, one wave each
//   kind 1  band `index & 31`: workgroup 0 of the group luma, workgroup 1 chroma; a deblocking band (deblock_device.h) -- or, index bit 14 set and only in
//           the k_chain_i variant (chain_intra.hip), a band of the intra wavefront of a picture that is (mostly) intra coded
// The host orders the work list (Engine::launch).  The dispatcher starts workgroups strictly in index order and a workgroup that waits stays
// resident, so the order decides what can run:
//   * all deblocking bands of the launch come first and are resident from the start (Engine::form keeps their number at half of what the GPU
//     holds at 3 workgroups per CU); a band lives for its whole wavefront and waits for reconstruction bits that are produced later;
//   * reconstruction groups follow along the pipeline's own time axis: the deblocking wavefront reaches macroblock (x, y) in step x + 2y, so the
//     segment gets the key  base(picture) + 2 * row + 8 * segment,  base growing by `lag` steps from picture to picture of a stream.  A
//     reconstruction wave waits for bands (resident) whose own needs have smaller keys, so the waiting group with the smallest key can always
//     run: no deadlock at any occupancy.  Other orders fill the machine with waiting workgroups: picture-major order held picture n+2 back until
//     all of picture n+1 was resident (0.5 ms between the pictures of a 1080p chain), row-major order inside a picture did the same to the
//     lower half of the diagonal wavefront (0.27 ms); 32-macroblock segments with a lag of 24 steps broke the key rule and deadlocked at 4 streams.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "jobs.h"
#include "kernels.h"
#include "kernel_common.h"
#include "chain_common.h"
#include "recon_device.h"
#include "deblock_device.h"

namespace jmamd {

template <int DEPTH>
__global__ __launch_bounds__(256) void k_chain(const PicParams *pics, int *ctl, int *err, const uint32_t *groups, int pub) {
    const int g = (int)blockIdx.x >> 1, rem = (int)blockIdx.x & 1;
    const uint32_t entry = groups[g];
    const PicParams &pp = pics[entry >> 16];
    const bool census_on = pub > 0 && (pub & 0x10000) != 0;
    if (pub > 0) pub &= 0xffff;
    ChainView cv{ctl, err}; cv.census_on = census_on;
    // ONE LDS block for whichever role the workgroup has: two static arrays add up (52.5 KB: three workgroups per CU by LDS alone), and since round 4
    // the kernel's 119 registers allow four
    __shared__ __align__(16) uint8_t smem[kDeblockSmemBytes > (int)sizeof(ReconLds) ? kDeblockSmemBytes : (int)sizeof(ReconLds)];
    cv.census(ChainView::CENSUS_MAX_GROUP, g);
    if (!(entry & 0x8000u)) {
        const int row = (int)(entry & 0x7fffu) >> 5, seg = (int)entry & 31;
        const int x = seg * 8 + rem * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform, and known to be (recon_device.h)
        const bool valid = row < pp.mb_h && x < pp.mb_w;
        if (row >= pp.mb_h || seg * 8 + rem * 4 >= pp.mb_w) return;
        cv.census(ChainView::CENSUS_RECON_STARTED); cv.stamp(pp.chain_idx, ChainView::STAMP_RECON_FIRST);
        const uint32_t tr0 = census_on ? (uint32_t)wall_clock64() : 0u;
        ReconLds &sm = *reinterpret_cast<ReconLds *>(smem);
        const int mb = valid ? row * pp.mb_w + x : 0;
        // one instantiation for every picture of the launch (with the cached-load variant beside it the kernel needs 196 VGPRs and scratch;
        // this way 165): a picture without references inside the launch passes wait_final at once
        recon_inter_wave<true, true, true, false>(pp, mb, valid, sm, cv);
        cv.census(ChainView::CENSUS_RECON_DONE); cv.stamp(pp.chain_idx, ChainView::STAMP_RECON_LAST);
        if (census_on) cv.census(ChainView::CENSUS_RECON_TICKS, (int)((uint32_t)wall_clock64() - tr0));
    } else {
        cv.census(ChainView::CENSUS_BAND_STARTED); cv.stamp(pp.chain_idx, ChainView::STAMP_BAND_FIRST);
        int *cpic = cv.pic(pp.chain_idx);
        deblock_band_body<DEPTH, true>(pp, (int)(entry & 31u), rem == 1, cpic + kChainRing, pub, smem, cpic, err + pp.chain_idx);
        cv.census(ChainView::CENSUS_BAND_DONE); cv.stamp(pp.chain_idx, ChainView::STAMP_BAND_LAST);
    }
}

bool chain_supported(int mb_w, int mb_h) {
    static const bool off = getenv("JM_AMD_DEC_NO_CHAIN") != nullptr;
    return !off && mb_w > 0 && mb_w <= 32 * kChainRowWords && mb_h <= kChainMaxRows && deblock_lds_supported(mb_w, mb_h);
}
int chain_ctl_ints() { return kChainStride; }
int chain_tail_ints() { return kChainTail; }
int chain_tail_head_ints() { return kChainTailHead; }
int chain_tail_wait_limit() { return kTailWaitLimit; }

int deblock_depth(); int deblock_pub();

int chain_band_rows() { return kBandRows; }

// chain_intra.hip: the variant with the intra role (its own translation unit: the inlining decisions, and with them the register count, of a fused
// kernel depend on everything else in the module)
void launch_chain_intra(const PicParams *d_pics, const uint32_t *d_groups, int n_groups, int *ctl, int *err, int depth, int pub, hipStream_t st);

int chain_intra_resident_workgroups();
// workgroups of a chain launch the current device holds at once: k_chain, or (intra) k_chain_i; 0 when the runtime cannot say
int chain_resident_workgroups(bool intra) {
    if (intra) return chain_intra_resident_workgroups();
    int per_cu = 0, dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    const int depth = deblock_depth();
    hipError_t e = depth <= 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chain<2>, 256, 0)
                 : depth == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chain<3>, 256,
                     0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chain<4>, 256, 0);
    return e == hipSuccess ? per_cu * prop.multiProcessorCount : 0;
}

void launch_chain(const PicParams *d_pics, const uint32_t *d_groups, int n_groups, bool with_intra, int *ctl, int *err, bool debug_stall, hipStream_t st) {
    static const bool census = getenv("JM_AMD_DEC_CENSUS") != nullptr;       // diagnostic launches count their workgroups (chain_common.h)
    const int depth = deblock_depth(), pub = debug_stall ? -1 : (deblock_pub() | (census ? 0x10000 : 0));
    if (with_intra) { launch_chain_intra(d_pics, d_groups, n_groups, ctl, err, depth, pub, st); return; }
    dim3 grid((unsigned)n_groups * 2u), block(256);
    if (depth <= 2) hipLaunchKernelGGL((k_chain<2>), grid, block, 0, st, d_pics, ctl, err, d_groups, pub);
    else if (depth == 3) hipLaunchKernelGGL((k_chain<3>), grid, block, 0, st, d_pics, ctl, err, d_groups, pub);
    else hipLaunchKernelGGL((k_chain<4>), grid, block, 0, st, d_pics, ctl, err, d_groups, pub);
}

}  // namespace jmamd
