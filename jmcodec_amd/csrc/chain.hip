// jmcodec_amd/csrc/chain.hip -- k_chain: motion compensation + residual AND in-loop deblocking of a batch of P / B pictures in ONE launch,
// with consecutive pictures of the same stream pipelined at macroblock granularity (protocol: chain_common.h).
//
// Part of the replacement for cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:33-41): the reference decodes one picture at a time on
// a fixed-function pipeline; here a single stream keeps several pictures in flight on the CUs, because picture n+1's macroblock (x, y) only
// needs picture n deblocked a few macroblocks beyond (x, y), not the whole picture.
//
// grid = (blocks per picture, pictures).  Workgroup roles by blockIdx.x, in dispatch order:
//   [0, n_recon)            reconstruction, 4 macroblocks per workgroup (one wave each), macroblock rows top to bottom so that workgroups
//                           retire in the order the deblocking wavefront of the reference picture releases them; inside a row the 8 XCDs
//                           (workgroup i runs on XCD i mod 8) each take one vertical strip of the picture, so reference windows of
//                           neighbouring macroblocks meet in one memory channel group (reference loads bypass the L2: see chain_common.h)
//   [n_recon, n_recon + 2 * bands)   deblocking bands (deblock_device.h), luma and chroma
// A workgroup only waits for lower block indices: reconstruction for deblocking bands of EARLIER pictures, bands for reconstruction bits of
// their own picture and for the band above.
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
__global__ __launch_bounds__(256) void k_chain(const PicParams *pics, int *ctl, int *err, int n_recon, int blocks_per_row, int pub) {
    const PicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & PS_CHAIN)) return;
    const ChainView cv{ctl, err};
    const int b = (int)blockIdx.x;
    if (b < n_recon) {
        const int row = b / blocks_per_row, rem = b - row * blocks_per_row;
        const int strip_w = (pp.mb_w + 7) >> 3;                               // macroblocks per XCD strip of THIS picture
        const int j = rem >> 3, xcd = rem & 7, in_strip = j * 4 + (int)(threadIdx.x >> 6);
        const int x = xcd * strip_w + in_strip;
        const bool valid = row < pp.mb_h && in_strip < strip_w && x < pp.mb_w;
        if (row >= pp.mb_h) return;
        __shared__ ReconLds sm;
        const int mb = valid ? row * pp.mb_w + x : 0;
        // one instantiation for every picture of the launch (with the cached-load variant beside it the kernel needs 196 VGPRs and scratch;
        // this way 165): a picture without references inside the launch passes wait_final at once
        recon_inter_wave<true, true>(pp, mb, valid, sm, cv);
    } else {
        const int k = b - n_recon;
        __shared__ __align__(16) uint8_t smem[kDeblockSmemBytes];
        int *cpic = cv.pic(pp.chain_idx);
        deblock_band_body<DEPTH, true>(pp, k >> 1, k & 1, cpic + kChainRing, pub, smem, cpic, err + pp.chain_idx);
    }
}

bool chain_supported(int mb_w, int mb_h) {
    static const bool off = getenv("JM_AMD_DEC_NO_CHAIN") != nullptr;
    return !off && mb_w > 0 && mb_w <= 32 * kChainRowWords && mb_h <= kChainMaxRows && deblock_lds_supported(mb_w, mb_h);
}
int chain_ctl_ints() { return kChainStride; }

int deblock_depth(); int deblock_pub();

void launch_chain(const PicParams *d_pics, int n, int max_mb_w, int max_mb_h, int *ctl, int *err, hipStream_t st) {
    const int strip_w = (max_mb_w + 7) / 8, blocks_per_row = 8 * ((strip_w + 3) / 4);
    const int n_recon = blocks_per_row * max_mb_h;                            // multiple of 8: block index mod 8 == XCD for every picture
    const int bands = (max_mb_h + kBandRows - 1) / kBandRows;
    const int gx = (n_recon + 2 * bands + 7) & ~7;
    const int depth = deblock_depth(), pub = deblock_pub();
    dim3 grid(gx, n), block(256);
    if (depth <= 2) hipLaunchKernelGGL((k_chain<2>), grid, block, 0, st, d_pics, ctl, err, n_recon, blocks_per_row, pub);
    else if (depth == 3) hipLaunchKernelGGL((k_chain<3>), grid, block, 0, st, d_pics, ctl, err, n_recon, blocks_per_row, pub);
    else hipLaunchKernelGGL((k_chain<4>), grid, block, 0, st, d_pics, ctl, err, n_recon, blocks_per_row, pub);
}

}  // namespace jmamd
