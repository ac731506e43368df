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
//   k_deblock_lds    ONE workgroup (16 waves) walks the steps in lockstep, one __syncthreads() per step.
//                    Waves 0-7 filter luma, waves 8-15 chroma; 16 lanes per macroblock (lane = pixel row
//                    for vertical edges, = pixel column for horizontal edges), so a whole edge chain of a
//                    macroblock stays in registers.  Everything a neighbour still needs lives in LDS:
//                      tile[row][x&1]   the macroblock's 16 rows after its own filtering (right 4 columns are
//                                       the next macroblock's left border)
//                      ring[row][x&3]   its bottom 4 rows (the macroblock below filters and finally stores them)
//                    HBM traffic is the algorithmic minimum: every sample is read once (prefetched one step
//                    ahead) and written once, as whole 16-byte row segments of a (-4,-4)-shifted 16x16 block.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "jobs.h"
#include "kernels.h"
#include "kernel_common.h"

namespace jmamd {

struct DbRec {                 // 96 bytes per macroblock: 48 for the luma workgroup, 48 for the chroma workgroup
    uint8_t y_bs[32];          // index dir*16 + edge*4 + segment : bS | tC0 << 3   (dir 0 = vertical edges)
    uint8_t y_ab[3][2];        // alpha, beta of [0 left MB edge, 1 internal edges, 2 top MB edge]
    uint8_t y_pad[10];
    uint8_t c_bs[2][16];       // [plane][dir*8 + (edge/2)*4 + segment] : bS | tC0 << 3
    uint8_t c_ab[2][3][2];     // [plane][class][alpha, beta]
    uint8_t c_pad[4];
};
static_assert(sizeof(DbRec) == 96, "DbRec must be 96 bytes");

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_deblock_prep(const PicParams *pics) {
    const PicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & PS_DEBLOCK_LDS)) return;
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
        bs = have ? boundary_strength(pp, pn, dir == 0 ? k * 4 + 3 : 12 + k, q, rq, true) : 0;
    } else bs = boundary_strength(pp, q, dir == 0 ? rq - 1 : rq - 4, q, rq, false);
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
// sample filters on register arrays
// ------------------------------------------------------------------------------------------
// s[0..7] = p3 p2 p1 p0 q0 q1 q2 q3
__device__ __forceinline__ void flt_luma(int *s, int bS, int alpha, int beta, int tc0) {
    int p3 = s[0], p2 = s[1], p1 = s[2], p0 = s[3], q0 = s[4], q1 = s[5], q2 = s[6], q3 = s[7];
    if (!(iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)) return;
    int ap = iabs(p2 - p0) < beta, aq = iabs(q2 - q0) < beta;
    if (bS < 4) {
        int tc = tc0 + ap + aq;
        int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        s[3] = clip1(p0 + delta); s[4] = clip1(q0 - delta);
        if (ap) s[2] = p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1);
        if (aq) s[5] = q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1);
    } else {
        bool strong = iabs(p0 - q0) < ((alpha >> 2) + 2);
        if (ap && strong) { s[3] = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3; s[2] = (p2 + p1 + p0 + q0 + 2) >> 2; s[1] = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3; }
        else s[3] = (2 * p1 + p0 + q1 + 2) >> 2;
        if (aq && strong) { s[4] = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3; s[5] = (p0 + q0 + q1 + q2 + 2) >> 2; s[6] = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3; }
        else s[4] = (2 * q1 + q0 + p1 + 2) >> 2;
    }
}
// chroma: p1 p0 q0 q1 by reference
__device__ __forceinline__ void flt_chroma(int p1, int &p0, int &q0, int q1, int bS, int alpha, int beta, int tc0) {
    if (!(iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)) return;
    if (bS < 4) {
        int tc = tc0 + 1;
        int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        p0 = clip1(p0 + delta); q0 = clip1(q0 - delta);
    } else { int np = (2 * p1 + p0 + q1 + 2) >> 2, nq = (2 * q1 + q0 + p1 + 2) >> 2; p0 = np; q0 = nq; }
}

__device__ __forceinline__ uint32_t pack4(const int *v) { return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24); }

// LDS layout (dynamic): per macroblock row
//   lumaTile  [2][16][16]   = 512 B        chromaTile [2][8][16] = 256 B
//   lumaRing  [4][4][16]    = 256 B        chromaRing [4][2][16] = 128 B
// then 64 x 64 B staging for the DbRec of the macroblock each group is working on.
constexpr int kGroups = 32;              // macroblock rows in flight per plane type: 8 waves x 4 groups
constexpr int kMaxSlots = 5;             // rows g, g+32, ... g+128 -> pictures up to 160 MB rows (2560 lines)

// Workgroup 0 (luma) and workgroup 1 (chroma) each own a private LDS image laid out the same way:
// 32 x 64 B DbRec staging, then per macroblock row the tile pair, then per row the ring.
struct Lds {
    // Row strides are padded by 16 bytes (4 banks): the four macroblock rows one wave works on would otherwise sit exactly
    // 512 / 256 / 128 bytes apart, i.e. in the same LDS banks, and every byte-column access would be a 4-way bank conflict.
    uint8_t *base; int mb_h; int hdr;     // hdr = bytes of DbRec staging in front (kRecStride per group)
    static constexpr int kRecStride = 80, kLT = 528, kLR = 272, kCT = 272, kCR = 144;
    __device__ uint8_t *rec(int group) const { return base + group * kRecStride; }
    __device__ uint8_t *luma_tile(int row, int par) const { return base + hdr + (size_t)row * kLT + par * 256; }
    __device__ uint8_t *luma_ring(int row, int slot) const { return base + hdr + (size_t)mb_h * kLT + (size_t)row * kLR + slot * 64; }
    __device__ uint8_t *chroma_tile(int row, int par) const { return base + hdr + (size_t)row * kCT + par * 128; }
    __device__ uint8_t *chroma_ring(int row, int slot) const { return base + hdr + (size_t)mb_h * kCT + (size_t)row * kCR + slot * 32; }
};

// ------------------------------------------------------------------------------------------
// luma: one macroblock, 16 lanes (l = 0..15)
// ------------------------------------------------------------------------------------------
__device__ void luma_mb(const PicParams &pp, const Lds &lds, int x, int row, int l, int group, uint4 own, uint32_t recdw) {
    uint8_t *tc = lds.luma_tile(row, x & 1), *tp = lds.luma_tile(row, (x - 1) & 1);
    uint8_t *ring_up = row > 0 ? lds.luma_ring(row - 1, x & 3) : nullptr;
    uint8_t *ring_up_l = row > 0 ? lds.luma_ring(row - 1, (x - 1) & 3) : nullptr;
    uint8_t *ring_dn = lds.luma_ring(row, x & 3), *ring_dn_l = lds.luma_ring(row, (x - 1) & 3);
    uint8_t *rec = lds.rec(group);
    if (l < 12) ((uint32_t *)rec)[l] = recdw;                 // 48-byte luma half of the DbRec
    // this lane's four vertical-edge and four horizontal-edge strengths (bS | tC0 << 3), and the class parameters
    int vb[4], hb[4], ab[6];
#pragma unroll
    for (int e = 0; e < 4; e++) { vb[e] = rec[e * 4 + (l >> 2)]; hb[e] = rec[16 + e * 4 + (l >> 2)]; }
#pragma unroll
    for (int i = 0; i < 6; i++) ab[i] = rec[32 + i];
    // ---- vertical edges: lane = pixel row l ----
    uint32_t left = x > 0 ? *(const uint32_t *)(tp + l * 16 + 12) : 0;
    int p[20];
    { uint32_t w[5] = {left, own.x, own.y, own.z, own.w};
#pragma unroll
      for (int i = 0; i < 20; i++) p[i] = (w[i >> 2] >> ((i & 3) * 8)) & 255; }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        int bs = vb[e] & 7;
        if (bs) { const int c = e ? 1 : 0; flt_luma(p + 4 * e, bs, ab[2 * c], ab[2 * c + 1], vb[e] >> 3); }
    }
    uint32_t left_after = pack4(p);
    if (x > 0) *(uint32_t *)(tp + l * 16 + 12) = left_after;
    *(uint4 *)(tc + l * 16) = make_uint4(pack4(p + 4), pack4(p + 8), pack4(p + 12), pack4(p + 16));
    // the left neighbour's bottom rows (its ring slot) get our edge-0 result for columns 12..15
    if (x > 0 && l >= 12) *(uint32_t *)(ring_dn_l + (l - 12) * 16 + 12) = left_after;
    // ---- horizontal edges: lane = pixel column l ----
    int c[20];
#pragma unroll
    for (int j = 0; j < 4; j++) c[j] = ring_up ? ring_up[j * 16 + l] : 0;
#pragma unroll
    for (int j = 0; j < 16; j++) c[4 + j] = tc[j * 16 + l];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        int bs = hb[e] & 7;
        if (bs) { const int k = e ? 1 : 2; flt_luma(c + 4 * e, bs, ab[2 * k], ab[2 * k + 1], hb[e] >> 3); }
    }
    if (ring_up) {
#pragma unroll
        for (int j = 1; j < 4; j++) ring_up[j * 16 + l] = (uint8_t)c[j];
    }
#pragma unroll
    for (int j = 0; j < 12; j++) tc[j * 16 + l] = (uint8_t)c[4 + j];
#pragma unroll
    for (int j = 12; j < 16; j++) { tc[j * 16 + l] = (uint8_t)c[4 + j]; ring_dn[(j - 12) * 16 + l] = (uint8_t)c[4 + j]; }
    // ---- store the final (-4,-4)-shifted 16x16 block: lane -> row R = l - 4 ----
    uint8_t *dst = pp.surf[pp.cur];
    int x0 = x * 16, y0 = row * 16, pitch = pp.pitch;
    bool right = x == pp.mb_w - 1, bottom = row == pp.mb_h - 1;
    {
        int R = l - 4;
        const uint8_t *src, *srcl;
        if (R < 0) { src = ring_up ? ring_up + (R + 4) * 16 : nullptr; srcl = ring_up_l ? ring_up_l + (R + 4) * 16 + 12 : nullptr; }
        else { src = tc + R * 16; srcl = tp + R * 16 + 12; }
        if (src) {
            uint8_t *d = dst + (size_t)(y0 + R) * pitch + x0;
            uint4 v = *(const uint4 *)src;
            if (x > 0) { uint32_t lf = *(const uint32_t *)srcl; *(uint4 *)(d - 4) = make_uint4(lf, v.x, v.y, v.z); }
            else { *(uint2 *)d = make_uint2(v.x, v.y); *(uint32_t *)(d + 8) = v.z; }
            if (right) *(uint32_t *)(d + 12) = v.w;
        }
    }
    if (bottom && l < 4) {
        int R = 12 + l;
        uint8_t *d = dst + (size_t)(y0 + R) * pitch + x0;
        uint4 v = *(const uint4 *)(tc + R * 16);
        if (x > 0) { uint32_t lf = *(const uint32_t *)(tp + R * 16 + 12); *(uint4 *)(d - 4) = make_uint4(lf, v.x, v.y, v.z); }
        else { *(uint2 *)d = make_uint2(v.x, v.y); *(uint32_t *)(d + 8) = v.z; }
        if (right) *(uint32_t *)(d + 12) = v.w;
    }
}

// ------------------------------------------------------------------------------------------
// chroma (NV12 interleaved UV): one macroblock, 16 lanes
// ------------------------------------------------------------------------------------------
__device__ void chroma_mb(const PicParams &pp, const Lds &lds, int x, int row, int l, int group, uint4 own, uint32_t recdw) {
    uint8_t *tc = lds.chroma_tile(row, x & 1), *tp = lds.chroma_tile(row, (x - 1) & 1);
    uint8_t *ring_up = row > 0 ? lds.chroma_ring(row - 1, x & 3) : nullptr;
    uint8_t *ring_up_l = row > 0 ? lds.chroma_ring(row - 1, (x - 1) & 3) : nullptr;
    uint8_t *ring_dn = lds.chroma_ring(row, x & 3), *ring_dn_l = lds.chroma_ring(row, (x - 1) & 3);
    uint8_t *rec = lds.rec(group);
    if (l < 12) ((uint32_t *)rec)[l] = recdw;                 // 48-byte chroma half of the DbRec
    // ---- vertical edges (chroma columns 0 and 4 <-> luma edges 0 and 2): lane = (plane, chroma row) ----
    {
        int plane = l >> 3, r = l & 7;
        // `own` of lane l is row (l & 7) of the macroblock (both lane halves prefetch the same 16 bytes)
        uint32_t left = x > 0 ? *(const uint32_t *)(tp + r * 16 + 12) : 0;
        if (plane == 0) *(uint4 *)(tc + r * 16) = own;
        uint32_t w[5] = {left, own.x, own.y, own.z, own.w};
        int b[20];
#pragma unroll
        for (int i = 0; i < 20; i++) b[i] = (w[i >> 2] >> ((i & 3) * 8)) & 255;
        int vb[2] = { rec[plane * 16 + (r >> 1)], rec[plane * 16 + 4 + (r >> 1)] };
        int ab[4] = { rec[32 + plane * 6], rec[32 + plane * 6 + 1], rec[32 + plane * 6 + 2], rec[32 + plane * 6 + 3] };
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            int bs = vb[e >> 1] & 7;
            if (!bs) continue;
            const int k = e ? 1 : 0;
            int o = 4 * e + plane;                                        // p1 = b[o], p0 = b[o+2], q0 = b[o+4], q1 = b[o+6]
            int p0 = b[o + 2], q0 = b[o + 4];
            flt_chroma(b[o], p0, q0, b[o + 6], bs, ab[2 * k], ab[2 * k + 1], vb[e >> 1] >> 3);
            if (e == 0) { tp[r * 16 + 14 + plane] = (uint8_t)p0; tc[r * 16 + plane] = (uint8_t)q0; }
            else { tc[r * 16 + 6 + plane] = (uint8_t)p0; tc[r * 16 + 8 + plane] = (uint8_t)q0; }
        }
    }
    // left neighbour's bottom rows: columns 12..15 (bytes) of rows 6, 7 after our edge 0
    if (x > 0 && l < 2) *(uint32_t *)(ring_dn_l + l * 16 + 12) = *(const uint32_t *)(tp + (6 + l) * 16 + 12);
    // ---- horizontal edges (chroma rows 0 and 4): lane = interleaved byte column ----
    {
        int plane = l & 1;
        int hb[2] = { rec[plane * 16 + 8 + (l >> 2)], rec[plane * 16 + 12 + (l >> 2)] };
        int ab[4] = { rec[32 + plane * 6 + 4], rec[32 + plane * 6 + 5], rec[32 + plane * 6 + 2], rec[32 + plane * 6 + 3] };   // top class, internal class
        int c[10];
        c[0] = ring_up ? ring_up[l] : 0; c[1] = ring_up ? ring_up[16 + l] : 0;
#pragma unroll
        for (int j = 0; j < 8; j++) c[2 + j] = tc[j * 16 + l];
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            int bs = hb[e >> 1] & 7;
            if (!bs) continue;
            const int k = e ? 1 : 0;                                      // ab[0..1] top class, ab[2..3] internal
            const int o = 2 * e;                                          // p1 = c[o], p0 = c[o+1], q0 = c[o+2], q1 = c[o+3]
            flt_chroma(c[o], c[o + 1], c[o + 2], c[o + 3], bs, ab[2 * k], ab[2 * k + 1], hb[e >> 1] >> 3);
        }
        if (ring_up) ring_up[16 + l] = (uint8_t)c[1];
        tc[l] = (uint8_t)c[2]; tc[3 * 16 + l] = (uint8_t)c[5]; tc[4 * 16 + l] = (uint8_t)c[6];
        ring_dn[l] = (uint8_t)c[8]; ring_dn[16 + l] = (uint8_t)c[9];
    }
    // ---- store the (-2 px, -2 rows)-shifted 8 x 16-byte block: lanes 0..7 -> row R = l - 2 ----
    uint8_t *dst = pp.surf[pp.cur] + pp.chroma_offset;
    int x0 = x * 16, y0 = row * 8, pitch = pp.pitch;
    bool right = x == pp.mb_w - 1, bottom = row == pp.mb_h - 1;
    if (l < 8) {
        int R = l - 2;
        const uint8_t *src, *srcl;
        if (R < 0) { src = ring_up ? ring_up + (R + 2) * 16 : nullptr; srcl = ring_up_l ? ring_up_l + (R + 2) * 16 + 12 : nullptr; }
        else { src = tc + R * 16; srcl = tp + R * 16 + 12; }
        if (src) {
            uint8_t *d = dst + (size_t)(y0 + R) * pitch + x0;
            uint4 v = *(const uint4 *)src;
            if (x > 0) { uint32_t lf = *(const uint32_t *)srcl; *(uint4 *)(d - 4) = make_uint4(lf, v.x, v.y, v.z); }
            else { *(uint2 *)d = make_uint2(v.x, v.y); *(uint32_t *)(d + 8) = v.z; }
            if (right) *(uint32_t *)(d + 12) = v.w;
        }
    } else if (bottom && l < 10) {
        int R = 6 + (l - 8);
        uint8_t *d = dst + (size_t)(y0 + R) * pitch + x0;
        uint4 v = *(const uint4 *)(tc + R * 16);
        if (x > 0) { uint32_t lf = *(const uint32_t *)(tp + R * 16 + 12); *(uint4 *)(d - 4) = make_uint4(lf, v.x, v.y, v.z); }
        else { *(uint2 *)d = make_uint2(v.x, v.y); *(uint32_t *)(d + 8) = v.z; }
        if (right) *(uint32_t *)(d + 12) = v.w;
    }
}

// ------------------------------------------------------------------------------------------
// grid = 2 workgroups of 512 threads: block 0 filters luma, block 1 chroma (independent planes, no exchange).
// 512 threads = 2 waves per SIMD, so each lane may use up to 256 VGPRs: the edge chains never spill.
// GROUPS 16-lane groups per workgroup (one macroblock row each per slot), NSLOTS rows per group.  (64, 3): 1024 threads = 4 waves
// per SIMD hide the latency of the dependent filter chains, and at most one macroblock per group is active in a step for pictures
// up to 64 rows tall / two up to 128 (instead of two / three with 32 groups); needs <= 128 VGPRs.  (32, 5): the 512-thread form.
__device__ int g_dbg_noload = 0;     // experiment: skip the per-step global loads (timing only, output is wrong)
// DEPTH = how many steps ahead samples and records are fetched: a step is shorter than a loaded HBM round trip, so with DEPTH 1 every
// step waited for its own prefetch (0.71 ms without the loads against 1.11 ms with them, 32 x 1080p).
template <int GROUPS, int NSLOTS, int DEPTH>
__global__ __launch_bounds__(GROUPS * 16) void k_deblock_lds(const PicParams *pics) {
    extern __shared__ __align__(16) uint8_t smem[];
    const PicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & PS_DEBLOCK_LDS)) return;
    const DbRec *recs = (const DbRec *)pp.dbrec;
    Lds lds{smem, pp.mb_h, GROUPS * Lds::kRecStride};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool is_chroma = blockIdx.x == 1;
    const int group = wave * 4 + (lane >> 4), l = lane & 15;
    const int mb_w = pp.mb_w, mb_h = pp.mb_h, pitch = pp.pitch;
    const uint8_t *plane_base = pp.surf[pp.cur] + (is_chroma ? pp.chroma_offset : 0);
    const int rows_per_mb = is_chroma ? 8 : 16;
    const int my_row = is_chroma ? (l & 7) : l;
    const int rec_dw = (is_chroma ? 12 : 0) + (l < 12 ? l : 0);
    uint4 pre_pix[DEPTH][NSLOTS]; uint32_t pre_rec[DEPTH][NSLOTS];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int k = 0; k < NSLOTS; k++) { pre_pix[d][k] = make_uint4(0, 0, 0, 0); pre_rec[d][k] = 0; }
    const int n_steps = mb_w + 2 * (mb_h - 1);
    const bool loads_on = !g_dbg_noload;
    // stage d holds what step (s + d) needs; fill stages 0 .. DEPTH-1 for steps 0 .. DEPTH-1
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int k = 0; k < NSLOTS; k++) {
            int row = group + GROUPS * k, xn = d - 2 * row;
            if (row < mb_h && xn >= 0 && xn < mb_w && loads_on) {
                pre_pix[d][k] = *(const uint4 *)(plane_base + (size_t)(row * rows_per_mb + my_row) * pitch + xn * 16);
                pre_rec[d][k] = ((const uint32_t *)&recs[row * mb_w + xn])[rec_dw];
            }
        }
    for (int s = 0; s < n_steps; s++) {
        // (1) take delivery of stage 0 (fetched DEPTH steps ago).  The empty asm "uses" the registers, so the compiler's s_waitcnt
        //     for exactly these loads lands HERE, before this step's loads are issued; younger stages stay in flight.
        uint4 own[NSLOTS]; uint32_t rdw[NSLOTS];
#pragma unroll
        for (int k = 0; k < NSLOTS; k++) {
            own[k] = pre_pix[0][k]; rdw[k] = pre_rec[0][k];
            asm volatile("" : "+v"(own[k].x), "+v"(own[k].y), "+v"(own[k].z), "+v"(own[k].w), "+v"(rdw[k]));
        }
        // (2) shift the stages and fetch for step s + DEPTH
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; d++)
#pragma unroll
            for (int k = 0; k < NSLOTS; k++) { pre_pix[d][k] = pre_pix[d + 1][k]; pre_rec[d][k] = pre_rec[d + 1][k]; }
#pragma unroll
        for (int k = 0; k < NSLOTS; k++) {
            int row = group + GROUPS * k;
            int xn = s + DEPTH - 2 * row;
            if (row < mb_h && xn >= 0 && xn < mb_w && loads_on) {
                pre_pix[DEPTH - 1][k] = *(const uint4 *)(plane_base + (size_t)(row * rows_per_mb + my_row) * pitch + xn * 16);
                pre_rec[DEPTH - 1][k] = ((const uint32_t *)&recs[row * mb_w + xn])[rec_dw];
            }
        }
        // (3) filter this step's macroblocks out of registers + LDS
#pragma unroll
        for (int k = 0; k < NSLOTS; k++) {
            int row = group + GROUPS * k;
            int x = s - 2 * row;
            if (row < mb_h && x >= 0 && x < mb_w) {
                if (is_chroma) chroma_mb(pp, lds, x, row, l, group, own[k], rdw[k]);
                else luma_mb(pp, lds, x, row, l, group, own[k], rdw[k]);
            }
        }
        // (4) step barrier: only LDS traffic has to be complete (a __syncthreads() would also drain vmcnt,
        //     i.e. wait for the prefetches that are in flight)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// ------------------------------------------------------------------------------------------
size_t deblock_lds_bytes(int mb_h) { return 64 * Lds::kRecStride + (size_t)mb_h * (Lds::kLT + Lds::kLR); }      // luma workgroup's need (chroma needs half)
bool deblock_lds_supported(int mb_w, int mb_h) { return mb_h <= kGroups * kMaxSlots && deblock_lds_bytes(mb_h) <= 160 * 1024 - 1024; }

void launch_deblock_lds(const PicParams *d_pics, int n, int max_mbs, int max_mb_h, hipStream_t st) {
    static bool attr_set[64] = {false};
    int dev = 0;
    hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        hipFuncSetAttribute((const void *)k_deblock_lds<32, 5, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        hipFuncSetAttribute((const void *)k_deblock_lds<64, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        hipFuncSetAttribute((const void *)k_deblock_lds<64, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        attr_set[dev] = true;
        if (getenv("JM_AMD_DEC_EXP_NOLOAD")) { int one = 1; hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_noload), &one, sizeof one); }
    }
    hipLaunchKernelGGL(k_deblock_prep, dim3(((max_mbs + 7) / 8 + 7) & ~7, n), dim3(256), 0, st, d_pics);   // multiple of 8 (XCD bands)
    static const bool wide = !getenv("JM_AMD_DEC_DEBLOCK_512");
    static const bool deep = !getenv("JM_AMD_DEC_DEBLOCK_DEPTH1");
    if (wide && deep && max_mb_h <= 128) hipLaunchKernelGGL((k_deblock_lds<64, 2, 2>), dim3(2, n), dim3(1024), deblock_lds_bytes(max_mb_h), st, d_pics);
    else if (wide) hipLaunchKernelGGL((k_deblock_lds<64, 3, 1>), dim3(2, n), dim3(1024), deblock_lds_bytes(max_mb_h), st, d_pics);
    else hipLaunchKernelGGL((k_deblock_lds<32, 5, 1>), dim3(2, n), dim3(512), deblock_lds_bytes(max_mb_h), st, d_pics);
}

}  // namespace jmamd
