// jmcodec_amd/csrc/deblock_lds.hip -- in-loop deblocking (H.264 8.7), LDS-resident lockstep wavefront.
//
// Part of the replacement for cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:33-41).
//
// Clause 8.7 filters macroblocks in raster order, per macroblock the vertical edges and then the
// horizontal edges, and every filter reads samples its predecessors already changed.  The exact
// dependencies are: MB(x,y) after MB(x-1,y), MB(x,y-1) and MB(x+1,y-1).  So all macroblocks with the same
// s = x + 2y are independent ("step s") and the steps must run in order.
//
//   k_deblock_prep   fully parallel: boundary strengths (8.7.2.1) and the alpha/beta/tC0 of every edge class
//                    of every macroblock -> one 64-byte DbRec.  All divergent, table-driven work lives here.
//   k_deblock_band   one 4-wave workgroup per band of 16 macroblock rows walks the steps of its rows in lockstep, one s_barrier per
//                    step; bands are chained downwards through step counters in device memory.  16 lanes per macroblock (lane =
//                    pixel row for vertical edges, = pixel column for horizontal edges), so a whole edge chain of a
//                    macroblock stays in registers.  Everything a neighbour still needs lives in LDS:
//                      tile[row][x&1]   the macroblock's 16 rows after its own filtering (right 4 columns are
//                                       the next macroblock's left border)
//                      ring[row][x&3]   its bottom 4 rows (the macroblock below filters and finally stores them)
//                    HBM traffic is the algorithmic minimum: every sample is read once (prefetched DEPTH steps
//                    ahead) and written once, as whole 16-byte row segments of a (-4,-4)-shifted 16x16 block.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <algorithm>
#include "jobs.h"
#include "kernels.h"
#include "kernel_common.h"
#include "deblock_device.h"    // DbRec, filters, LDS layout, deblock_band_body (shared with chain.hip)

namespace jmamd {

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_deblock_prep(const PicParams *pics) {
    const PicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & (PS_DEBLOCK_LDS | PS_CHAIN))) return;
    DbRec *out = (DbRec *)pp.dbrec;
    const int per_xcd = ((int)gridDim.x + 7) >> 3;         // XCD-aware: one contiguous band of macroblocks per XCD (see k_recon_inter)
    const int blk = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    __shared__ __align__(16) uint8_t stage[8][96];         // records are assembled in LDS and written as 16-byte stores
    int sub = threadIdx.x >> 5, t = threadIdx.x & 31;
    int mb = blk * 8 + sub;
    int n_mbs = pp.mb_w * pp.mb_h;
    if (mb >= n_mbs) return;
    int mbx = mb % pp.mb_w, mby = mb / pp.mb_w;
    const MbW q = load_mbw(&pp.mbs[mb]);
    const SliceRec sl = pp.slices[mbw_slice(q)];
    bool has_left = mbx > 0, has_top = mby > 0;
    MbW pl = q, pt = q;
    if (has_left) { pl = load_mbw(&pp.mbs[mb - 1]); if (sl.disable == 2 && mbw_slice(pl) != mbw_slice(q)) has_left = false; }
    if (has_top) { pt = load_mbw(&pp.mbs[mb - pp.mb_w]); if (sl.disable == 2 && mbw_slice(pt) != mbw_slice(q)) has_top = false; }
    // lane t: one boundary strength (8.7.2.1)
    int dir = t >> 4, e = (t >> 2) & 3, k = t & 3;
    int rq = dir == 0 ? k * 4 + e : e * 4 + k;
    int bs;
    if (sl.disable == 1) bs = 0;
    else if (e == 0) {
        bool have = dir == 0 ? has_left : has_top;
        const MbW pn = select_mbw(dir == 0, pl, pt);
        bs = have ? boundary_strength(pp, pn, dir == 0 ? k * 4 + 3 : 12 + k, q, rq, true, dir == 1) : 0;
    } else bs = boundary_strength(pp, q, dir == 0 ? rq - 1 : rq - 4, q, rq, false, dir == 1);
    if ((e & 1) && (mbw_modes(q) & MBM_T8X8)) bs = 0;      // 8.7: with the 8x8 transform only the 8x8 block edges are filtered (chroma uses e = 0, 2)
    // edge class of this lane: 0 left MB edge, 1 internal, 2 top MB edge; qPp is the neighbour's QP on MB edges
    int cls = e ? 1 : (dir ? 2 : 0);
    int qp_p = cls == 0 ? mbw_qp(pl) : (cls == 2 ? mbw_qp(pt) : mbw_qp(q)), qp_q = mbw_qp(q);
    DbRec *o = (DbRec *)stage[sub];
    {
        int qpav = (qp_p + qp_q + 1) >> 1;
        int ia = clip3(0, 51, qpav + sl.alpha_off), ib = clip3(0, 51, qpav + sl.beta_off);
        int tc0 = (bs >= 1 && bs <= 3) ? kTc0[ia][bs - 1] : 0;
        o->y_bs[t] = (uint8_t)(bs | (tc0 << 3));
        if (k == 0 && (e == 0 || (e == 1 && dir == 0))) { o->y_ab[cls][0] = kAlpha[ia]; o->y_ab[cls][1] = kBeta[ib]; }
    }
    if (!(e & 1)) {
#pragma unroll
        for (int plane = 0; plane < 2; plane++) {
            int off = plane ? pp.cr_qp_off : pp.cb_qp_off;
            int qpav = (chroma_qp(qp_p, off) + chroma_qp(qp_q, off) + 1) >> 1;
            int ia = clip3(0, 51, qpav + sl.alpha_off), ib = clip3(0, 51, qpav + sl.beta_off);
            int tc0 = (bs >= 1 && bs <= 3) ? kTc0[ia][bs - 1] : 0;
            o->c_bs[plane][dir * 8 + (e >> 1) * 4 + k] = (uint8_t)(bs | (tc0 << 3));
            if (k == 0 && (e == 0 || (e == 2 && dir == 0))) { o->c_ab[plane][cls][0] = kAlpha[ia]; o->c_ab[plane][cls][1] = kBeta[ib]; }
        }
    }
    // the 32 lanes of this macroblock are one half-wave: LDS writes above are visible to the reads below (same wave)
    if (t < 6) ((uint4 *)&out[mb])[t] = ((const uint4 *)stage[sub])[t];
}

// ------------------------------------------------------------------------------------------
// k_deblock_band: the stage kernel -- grid (2 x bands, pictures); ctl = the batch's control buffer (chain_common.h layout), err = its error words
template <int DEPTH>
__global__ __launch_bounds__(kBandRows * 16) void k_deblock_band(const PicParams *pics, int *ctl, int pub, int *err) {
    __shared__ __align__(16) uint8_t smem[kDeblockSmemBytes];
    const PicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & PS_DEBLOCK_LDS)) return;
    deblock_band_body<DEPTH, false>(pp, blockIdx.x >> 1, blockIdx.x & 1, ctl + (size_t)blockIdx.y * kChainStride + kChainRing, pub, smem, nullptr,
        err + blockIdx.y);
}

// ------------------------------------------------------------------------------------------
bool deblock_lds_supported(int mb_w, int mb_h) { return mb_w > 0 && mb_h <= kBandRows * kDeblockMaxBands; }

// prefetch depth of the band kernels (steps in flight): 2 measured 2 % faster than 3 (609-614 against 624-627 us per launch), 4 1 % slower
// (profiles/r04_ab7_chain_fixed.json); kDeblockPub: the band publishes its step counter every second step
constexpr int kDeblockDepth = 2, kDeblockPub = 2;
int deblock_depth() { return kDeblockDepth; }
int deblock_row_lag() { return kRowLag; }     // steps between macroblock rows of the deblocking wavefront (Engine::launch orders chain work lists by it)
int deblock_pub() { return kDeblockPub; }

void launch_deblock_prep(const PicParams *d_pics, int n, int max_mbs, hipStream_t st) {
    hipLaunchKernelGGL(k_deblock_prep, dim3(((max_mbs + 7) / 8 + 7) & ~7, n), dim3(256), 0, st, d_pics);   // multiple of 8 (XCD bands)
}
// ctl must have been cleared for this batch (Engine::launch does one memset for all H.264 kernels of a batch)
void launch_deblock_lds(const PicParams *d_pics, int n, int max_mb_h, int *ctl, int *err, bool debug_stall, hipStream_t st) {
    const int depth = deblock_depth(), pub = debug_stall ? -1 : deblock_pub();
    const int bands = (max_mb_h + kBandRows - 1) / kBandRows;
    dim3 grid(2 * bands, n), block(kBandRows * 16);
    if (depth <= 2) hipLaunchKernelGGL((k_deblock_band<2>), grid, block, 0, st, d_pics, ctl, pub, err);
    else if (depth == 3) hipLaunchKernelGGL((k_deblock_band<3>), grid, block, 0, st, d_pics, ctl, pub, err);
    else hipLaunchKernelGGL((k_deblock_band<4>), grid, block, 0, st, d_pics, ctl, pub, err);
}

}  // namespace jmamd
