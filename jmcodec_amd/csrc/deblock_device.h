// jmcodec_amd/csrc/deblock_device.h -- device code of the LDS-resident deblocking wavefront (H.264 8.7), shared by k_deblock_band
// (deblock_lds.hip: one launch per stage) and k_chain (chain.hip: consecutive pictures of a stream in one launch).  See deblock_lds.hip
// for the algorithm; part of the replacement for cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:33-41).
#pragma once
#include <hip/hip_runtime.h>
#include "jobs.h"
#include "kernels.h"
#include "kernel_common.h"
#include "chain_common.h"

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
// sample filters on register arrays
// ------------------------------------------------------------------------------------------
// s[0..7] = p3 p2 p1 p0 q0 q1 q2 q3; bsw = bS | tC0 << 3.  Written without divergent branches: the two filters of 8.7.2.3 / 8.7.2.4
// are computed for every lane and selected, and the only branches are wave-uniform (nothing to filter / no lane with bS = 4).  The
// branchy form cost ~10 exec-mask regions and, in the horizontal pass, ~110 register copies at the joins per edge.
// |a - b| of two samples (0..255, upper bytes zero): one v_sad_u8 instead of sub / neg / max
__device__ __forceinline__ int adiff(int a, int b) { return (int)__builtin_amdgcn_sad_u8((unsigned)a, (unsigned)b, 0u); }
__device__ __forceinline__ int sel(bool c, int a, int b) { return c ? a : b; }      // operands are evaluated by the caller: a v_cndmask, never a branch
__device__ __forceinline__ void flt_luma(int *s, int bsw, int alpha, int beta) {
    const int bS = bsw & 7, tc0 = bsw >> 3;
    const int p3 = s[0], p2 = s[1], p1 = s[2], p0 = s[3], q0 = s[4], q1 = s[5], q2 = s[6], q3 = s[7];
    const bool on = ((int)(bS != 0) & (int)(adiff(p0, q0) < alpha) & (int)(adiff(p1, p0) < beta) & (int)(adiff(q1, q0) < beta)) != 0;
    if (!__builtin_amdgcn_ballot_w64(on)) return;
    const bool ap = adiff(p2, p0) < beta, aq = adiff(q2, q0) < beta;
    const int tc = tc0 + (int)ap + (int)aq;
    const int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
    const int avg = (p0 + q0 + 1) >> 1;
    const bool nrm = ((int)on & (int)(bS < 4)) != 0;
    const int n_p0 = clip1(p0 + delta), n_q0 = clip1(q0 - delta);
    const int n_p1 = p1 + clip3(-tc0, tc0, (p2 + avg - (p1 << 1)) >> 1), n_q1 = q1 + clip3(-tc0, tc0, (q2 + avg - (q1 << 1)) >> 1);
    int r_p0 = sel(nrm, n_p0, p0), r_q0 = sel(nrm, n_q0, q0), r_p1 = sel(((int)nrm & (int)ap) != 0, n_p1, p1), r_q1 = sel(((int)nrm & (int)aq) != 0, n_q1, q1), r_p2 = p2, r_q2 = q2;
    const bool st = ((int)on & (int)(bS >= 4)) != 0;
    if (__builtin_amdgcn_ballot_w64(st)) {
        const bool strong = adiff(p0, q0) < ((alpha >> 2) + 2);
        const bool sp = ((int)st & (int)ap & (int)strong) != 0, sq = ((int)st & (int)aq & (int)strong) != 0;
        const int w_p0 = (2 * p1 + p0 + q1 + 2) >> 2, w_q0 = (2 * q1 + q0 + p1 + 2) >> 2;
        const int s_p0 = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3, s_p1 = (p2 + p1 + p0 + q0 + 2) >> 2, s_p2 = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
        const int s_q0 = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3, s_q1 = (p0 + q0 + q1 + q2 + 2) >> 2, s_q2 = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
        r_p0 = sel(sp, s_p0, sel(st, w_p0, r_p0)); r_p1 = sel(sp, s_p1, r_p1); r_p2 = sel(sp, s_p2, r_p2);
        r_q0 = sel(sq, s_q0, sel(st, w_q0, r_q0)); r_q1 = sel(sq, s_q1, r_q1); r_q2 = sel(sq, s_q2, r_q2);
    }
    s[1] = r_p2; s[2] = r_p1; s[3] = r_p0; s[4] = r_q0; s[5] = r_q1; s[6] = r_q2;
}
// chroma: p1 p0 q0 q1 by reference
__device__ __forceinline__ void flt_chroma(int p1, int &p0, int &q0, int q1, int bsw, int alpha, int beta) {
    const int bS = bsw & 7, tc = (bsw >> 3) + 1;
    const bool on = ((int)(bS != 0) & (int)(adiff(p0, q0) < alpha) & (int)(adiff(p1, p0) < beta) & (int)(adiff(q1, q0) < beta)) != 0;
    const int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
    const int n_p0 = clip1(p0 + delta), n_q0 = clip1(q0 - delta), w_p0 = (2 * p1 + p0 + q1 + 2) >> 2, w_q0 = (2 * q1 + q0 + p1 + 2) >> 2;
    p0 = sel(on, sel(bS < 4, n_p0, w_p0), p0); q0 = sel(on, sel(bS < 4, n_q0, w_q0), q0);
}

__device__ __forceinline__ uint32_t pack4(const int *v) { return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24); }

// LDS layout (dynamic): per macroblock row
//   lumaTile  [2][16][16]   = 512 B        chromaTile [2][8][16] = 256 B
//   lumaRing  [4][4][16]    = 256 B        chromaRing [4][2][16] = 128 B
// then 64 x 64 B staging for the DbRec of the macroblock each group is working on.
// Luma and chroma workgroups each own a private LDS image laid out the same way:
// DbRec staging per group, then per macroblock row the tile pair, then per row the ring.
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

// What a macroblock step needs of the picture, fetched ONCE per workgroup.  The surface pointer is a global (address space 1) pointer:
// a generic pointer read from PicParams makes every access a FLAT instruction, which counts on lgkmcnt as well as vmcnt, so each
// wait for an LDS result also waited for the sample prefetches in flight; and reading pp.* inside the step put two dependent
// global loads on the critical path of every step.
typedef __attribute__((address_space(1))) uint8_t gbyte;
struct DbCtx { gbyte *plane; int pitch, mb_w, mb_h; };
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
#define JM_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ T gload(const gbyte *p);
template <> __device__ __forceinline__ uint4 gload<uint4>(const gbyte *p) { v4u v = *(const JM_GLOBAL v4u *)p; return make_uint4(v.x, v.y, v.z, v.w); }
template <> __device__ __forceinline__ uint32_t gload<uint32_t>(const gbyte *p) { return *(const JM_GLOBAL uint32_t *)p; }
__device__ __forceinline__ void gstore(gbyte *p, uint4 v) { v4u t = {v.x, v.y, v.z, v.w}; *(JM_GLOBAL v4u *)p = t; }
__device__ __forceinline__ void gstore(gbyte *p, uint2 v) { v2u t = {v.x, v.y}; *(JM_GLOBAL v2u *)p = t; }
__device__ __forceinline__ void gstore(gbyte *p, uint32_t v) { *(JM_GLOBAL uint32_t *)p = v; }
// final samples.  WT (chain launches): write-through stores (chain_common.h), because the next reader -- the motion compensation of the following picture --
// may sit on another XCD and reads with cache-bypassing loads as soon as the band's `fin` counter covers the step (chain_common.h)
__device__ __forceinline__ void gstore_wt(gbyte *p, uint32_t v) { __hip_atomic_store((JM_GLOBAL uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool WT> __device__ __forceinline__ void fstore(gbyte *p, uint4 v) { if (WT) st_wt16((void *)p, v); else gstore(p, v); }
template <bool WT> __device__ __forceinline__ void fstore(gbyte *p, uint2 v) { if (WT) st_wt8((void *)p, v); else gstore(p, v); }
template <bool WT> __device__ __forceinline__ void fstore(gbyte *p, uint32_t v) { if (WT) gstore_wt(p, v); else gstore(p, v); }

// ------------------------------------------------------------------------------------------
// luma: one macroblock, 16 lanes (l = 0..15)
// ------------------------------------------------------------------------------------------
// row = macroblock row in the picture, lrow = the row's index in the workgroup's LDS image (== row when one workgroup holds the whole plane)
template <bool WT>
__device__ __forceinline__ void luma_mb(const DbCtx &pp, const Lds &lds, int x, int row, int lrow, int l, int group, uint4 own, uint32_t recdw) {
    uint8_t *tc = lds.luma_tile(lrow, x & 1), *tp = lds.luma_tile(lrow, (x - 1) & 1);
    uint8_t *ring_up = row > 0 ? lds.luma_ring(lrow - 1, x & 3) : nullptr;
    uint8_t *ring_up_l = row > 0 ? lds.luma_ring(lrow - 1, (x - 1) & 3) : nullptr;
    uint8_t *ring_dn = lds.luma_ring(lrow, x & 3), *ring_dn_l = lds.luma_ring(lrow, (x - 1) & 3);
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
    for (int e = 0; e < 4; e++) { const int c = e ? 1 : 0; flt_luma(p + 4 * e, vb[e], ab[2 * c], ab[2 * c + 1]); }
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
    for (int e = 0; e < 4; e++) { const int k = e ? 1 : 2; flt_luma(c + 4 * e, hb[e], ab[2 * k], ab[2 * k + 1]); }
    if (ring_up) {
#pragma unroll
        for (int j = 1; j < 4; j++) ring_up[j * 16 + l] = (uint8_t)c[j];
    }
#pragma unroll
    for (int j = 0; j < 12; j++) tc[j * 16 + l] = (uint8_t)c[4 + j];
#pragma unroll
    for (int j = 12; j < 16; j++) { tc[j * 16 + l] = (uint8_t)c[4 + j]; ring_dn[(j - 12) * 16 + l] = (uint8_t)c[4 + j]; }
    // ---- store the final (-4,-4)-shifted 16x16 block: lane -> row R = l - 4 ----
    gbyte *dst = pp.plane;
    int x0 = x * 16, y0 = row * 16, pitch = pp.pitch;
    bool right = x == pp.mb_w - 1, bottom = row == pp.mb_h - 1;
    {
        int R = l - 4;
        const uint8_t *src, *srcl;
        if (R < 0) { src = ring_up ? ring_up + (R + 4) * 16 : nullptr; srcl = ring_up_l ? ring_up_l + (R + 4) * 16 + 12 : nullptr; }
        else { src = tc + R * 16; srcl = tp + R * 16 + 12; }
        if (src) {
            gbyte *d = dst + (size_t)(y0 + R) * pitch + x0;
            uint4 v = *(const uint4 *)src;
            if (x > 0) { uint32_t lf = *(const uint32_t *)srcl; fstore<WT>(d - 4, make_uint4(lf, v.x, v.y, v.z)); }
            else { fstore<WT>(d, make_uint2(v.x, v.y)); fstore<WT>(d + 8, v.z); }
            if (right) fstore<WT>(d + 12, v.w);
        }
    }
    if (bottom && l < 4) {
        int R = 12 + l;
        gbyte *d = dst + (size_t)(y0 + R) * pitch + x0;
        uint4 v = *(const uint4 *)(tc + R * 16);
        if (x > 0) { uint32_t lf = *(const uint32_t *)(tp + R * 16 + 12); fstore<WT>(d - 4, make_uint4(lf, v.x, v.y, v.z)); }
        else { fstore<WT>(d, make_uint2(v.x, v.y)); fstore<WT>(d + 8, v.z); }
        if (right) fstore<WT>(d + 12, v.w);
    }
}

// ------------------------------------------------------------------------------------------
// chroma (NV12 interleaved UV): one macroblock, 16 lanes
// ------------------------------------------------------------------------------------------
template <bool WT>
__device__ __forceinline__ void chroma_mb(const DbCtx &pp, const Lds &lds, int x, int row, int lrow, int l, int group, uint4 own, uint32_t recdw) {
    uint8_t *tc = lds.chroma_tile(lrow, x & 1), *tp = lds.chroma_tile(lrow, (x - 1) & 1);
    uint8_t *ring_up = row > 0 ? lds.chroma_ring(lrow - 1, x & 3) : nullptr;
    uint8_t *ring_up_l = row > 0 ? lds.chroma_ring(lrow - 1, (x - 1) & 3) : nullptr;
    uint8_t *ring_dn = lds.chroma_ring(lrow, x & 3), *ring_dn_l = lds.chroma_ring(lrow, (x - 1) & 3);
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
            const int k = e ? 1 : 0;
            int o = 4 * e + plane;                                        // p1 = b[o], p0 = b[o+2], q0 = b[o+4], q1 = b[o+6]
            int p0 = b[o + 2], q0 = b[o + 4];
            flt_chroma(b[o], p0, q0, b[o + 6], vb[e >> 1], ab[2 * k], ab[2 * k + 1]);
            if (vb[e >> 1] & 7) {
                if (e == 0) { tp[r * 16 + 14 + plane] = (uint8_t)p0; tc[r * 16 + plane] = (uint8_t)q0; }
                else { tc[r * 16 + 6 + plane] = (uint8_t)p0; tc[r * 16 + 8 + plane] = (uint8_t)q0; }
            }
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
            const int k = e ? 1 : 0;                                      // ab[0..1] top class, ab[2..3] internal
            const int o = 2 * e;                                          // p1 = c[o], p0 = c[o+1], q0 = c[o+2], q1 = c[o+3]
            flt_chroma(c[o], c[o + 1], c[o + 2], c[o + 3], hb[e >> 1], ab[2 * k], ab[2 * k + 1]);
        }
        if (ring_up) ring_up[16 + l] = (uint8_t)c[1];
        tc[l] = (uint8_t)c[2]; tc[3 * 16 + l] = (uint8_t)c[5]; tc[4 * 16 + l] = (uint8_t)c[6];
        ring_dn[l] = (uint8_t)c[8]; ring_dn[16 + l] = (uint8_t)c[9];
    }
    // ---- store the (-2 px, -2 rows)-shifted 8 x 16-byte block: lanes 0..7 -> row R = l - 2 ----
    gbyte *dst = pp.plane;
    int x0 = x * 16, y0 = row * 8, pitch = pp.pitch;
    bool right = x == pp.mb_w - 1, bottom = row == pp.mb_h - 1;
    if (l < 8) {
        int R = l - 2;
        const uint8_t *src, *srcl;
        if (R < 0) { src = ring_up ? ring_up + (R + 2) * 16 : nullptr; srcl = ring_up_l ? ring_up_l + (R + 2) * 16 + 12 : nullptr; }
        else { src = tc + R * 16; srcl = tp + R * 16 + 12; }
        if (src) {
            gbyte *d = dst + (size_t)(y0 + R) * pitch + x0;
            uint4 v = *(const uint4 *)src;
            if (x > 0) { uint32_t lf = *(const uint32_t *)srcl; fstore<WT>(d - 4, make_uint4(lf, v.x, v.y, v.z)); }
            else { fstore<WT>(d, make_uint2(v.x, v.y)); fstore<WT>(d + 8, v.z); }
            if (right) fstore<WT>(d + 12, v.w);
        }
    } else if (bottom && l < 10) {
        int R = 6 + (l - 8);
        gbyte *d = dst + (size_t)(y0 + R) * pitch + x0;
        uint4 v = *(const uint4 *)(tc + R * 16);
        if (x > 0) { uint32_t lf = *(const uint32_t *)(tp + R * 16 + 12); fstore<WT>(d - 4, make_uint4(lf, v.x, v.y, v.z)); }
        else { fstore<WT>(d, make_uint2(v.x, v.y)); fstore<WT>(d + 8, v.z); }
        if (right) fstore<WT>(d + 12, v.w);
    }
}

// ------------------------------------------------------------------------------------------
// A plane is cut into bands of kBandRows macroblock rows, one 4-wave workgroup (one wave per SIMD) per band, all bands of all pictures
// resident at once (the first form of this kernel walked a whole plane with ONE 16-wave workgroup and needed 159 KB of LDS).  A band runs the same steps s = x + 2 * row for its own rows; the only
// coupling is downwards: the last row of a band hands the bottom four sample rows of each macroblock (after its own filtering, i.e. the
// state clause 8.7 prescribes when the macroblock below starts) to the first row of the next band.  They travel through the picture
// surface itself (the band below overwrites them with the final values afterwards) and a per-band step counter in device memory:
//     band b publishes "steps completed" every `pub` steps (release); band b+1 polls it (acquire) only when the value it last saw does not
//     cover the step it is about to prefetch for, so in steady state it runs a few steps behind without waiting.
// Workgroups of a lower band have the higher block index, so whatever a workgroup waits for has been dispatched before it.
constexpr int kBandRows = 16;
constexpr int kDeblockSmemBytes = kBandRows * Lds::kRecStride + (kBandRows + 1) * (Lds::kLT + Lds::kLR);
// One 256-thread workgroup deblocks band `band` of one plane of picture pp.  prog = the picture's band step counters of the ring-row hand-over
// (kDeblockProgressStride ints).  CHAIN (k_chain): `cpic` = the picture's block of the chain buffer; the unfiltered samples come from
// reconstruction waves of the same launch (wait for their bits, read with cache-bypassing loads), the final samples are written through
// and the band publishes in cpic[kChainFin ..] how many of its steps are final in memory.
// AFTER_INTRA (k_chain_i only): the picture's unfiltered samples come from an intra wavefront of the same launch (see wait_recon below)
template <int DEPTH, bool CHAIN, bool AFTER_INTRA = false>
__device__ __forceinline__ void deblock_band_body(const PicParams &pp, int band, bool is_chroma, int *prog_pic, int pub, uint8_t *smem, int *cpic, int *err_word) {
    const int mb_w = pp.mb_w, mb_h = pp.mb_h, pitch = pp.pitch;
    const int row0 = band * kBandRows;
    if (row0 >= mb_h) return;
    const int rows = min(kBandRows, mb_h - row0);
    int *prog = prog_pic + (is_chroma ? kDeblockMaxBands : 0);
    const gbyte *recs = (const gbyte *)pp.dbrec;
    Lds lds{smem, kBandRows + 1, kBandRows * Lds::kRecStride};
    const int group = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int lrow = group + 1;
    const bool active = group < rows;
    const int row = row0 + (active ? group : 0);                             // idle groups shadow row0 (loads stay in bounds, nothing is used)
    gbyte *plane = (gbyte *)(pp.surf[pp.cur] + (is_chroma ? pp.chroma_offset : 0));
    const DbCtx cx{plane, pitch, mb_w, mb_h};
    const int rows_per_mb = is_chroma ? 8 : 16, ring_rows = is_chroma ? 2 : 4;
    const int my_row = is_chroma ? (l & 7) : l;
    const int rec_dw = (is_chroma ? 12 : 0) + (l < 12 ? l : 0);
    const bool takes_ring = band > 0 && group == 0;                        // first row of a lower band: ring rows come from the band above
    const bool gives_ring = active && group == rows - 1 && row < mb_h - 1;  // last row of a band that has a band below
    const int s_begin = 2 * row0, s_end = mb_w - 1 + 2 * (row0 + rows - 1);
    int known = 0;                                                          // steps the band above is known to have completed
    int *abort_word = CHAIN ? cpic - (size_t)pp.chain_idx * kChainStride + (size_t)kChainMaxPics * kChainStride : nullptr;   // launch-wide: a wait of this launch gave up (chain_common.h)
    // Hand-over protocol without cache maintenance: the ring rows and the counter are written and read with agent-scope relaxed atomics
    // (write-through stores / loads that bypass the non-coherent cache levels), ordered by a plain s_waitcnt on the writer's side and by the
    // data dependency on the counter on the reader's side.  Acquire / release at agent scope would write back and invalidate the whole
    // L2 of the XCD on every poll (measured: 2.8 ms per launch with one release per step).
    auto wait_above = [&](int need) {
        if (band == 0 || threadIdx.x >= 64 || known >= need) return;        // wave 0 holds group 0
        int spins = 0; uint32_t t0 = 0;
        while ((known = __hip_atomic_load(&prog[band - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need && ++spins < kSpinLimit &&
               !(CHAIN && (wait_expired(spins, t0) || ((spins & 255) == 0 && ld_coh(abort_word))))) __builtin_amdgcn_s_sleep(8);
        if (known < need) {                                                  // the band above never got there: damaged, and SAID so
            if (l == 0) { report_wait_timeout(err_word, CHAIN_ERR_RING_TIMEOUT); if (CHAIN) st_coh(abort_word, 1); }
            known = 0x7fffffff;                                              // do not wait again
        }
        asm volatile("" ::: "memory");
    };
    // CHAIN: macroblocks of this group's row known to be reconstructed, counted from the left (bits of the row's bitmap words seen so far)
    int recon_known = (pp.stages & PS_RECON) ? 0x7fffffff : 0;               // PS_RECON beside PS_CHAIN: reconstructed by the stage kernel before this launch
    const uint32_t *bits_row = CHAIN ? (const uint32_t *)(cpic + kChainBits) + (size_t)row * kChainRowWords : nullptr;
    // PS_CHAIN_INTRA: the unfiltered samples come from the picture's intra wavefront (which itself waited for the reconstruction bits).  Intra prediction
    // reads UNFILTERED neighbours: macroblock (x, row) may be deblocked once the intra wavefront has passed (x + 1, row + 1), i.e. its step s + 3, in the band
    // that holds row + 1 and in this row's own band (counters `ifin`, published two steps late like `fin`).
    constexpr bool after_intra = CHAIN && AFTER_INTRA;
    int ifin_known0 = 0, ifin_known1 = 0;
    const int *ifin0 = after_intra ? cpic + kChainIntraFin + (is_chroma ? 32 : 0) + (row >> 4) : nullptr;
    const int *ifin1 = after_intra ? cpic + kChainIntraFin + (is_chroma ? 32 : 0) + (min(row + 1, mb_h - 1) >> 4) : nullptr;
    auto wait_recon = [&](int x) {              // before the unfiltered samples of macroblock (x, row) are fetched; uniform per 16-lane group
        if (!CHAIN) return;
        const bool want = active && x >= 0 && x < mb_w;
        bool ok;
        if (after_intra) { const int need = x + 2 * row + 4; ok = wait_counter(ifin0, ifin_known0, want, need, abort_word) && wait_counter(ifin1, ifin_known1, want, need, abort_word); }
        else ok = wait_row_bit(bits_row, recon_known, want, x, abort_word);
        if (!ok && (threadIdx.x & 63) == 0) { report_wait_timeout(err_word, after_intra ? CHAIN_ERR_IFIN_TIMEOUT : CHAIN_ERR_BITS_TIMEOUT); st_coh(abort_word, 1); }
    };
    const int ring_lanes = ring_rows * 4;                                   // one dword per lane: ring row l >> 2, dword l & 3
    const int ring_lane = l < ring_lanes ? l : 0;
    const gbyte *pix_base = plane + (size_t)(row * rows_per_mb + my_row) * pitch;
    const gbyte *rec_base = recs + (size_t)row * mb_w * sizeof(DbRec) + rec_dw * 4;
    const gbyte *ring_base = plane + (size_t)(takes_ring ? row * rows_per_mb - ring_rows + (ring_lane >> 2) : 0) * pitch + (ring_lane & 3) * 4;
    // The loads of a stage are unconditional (clamped coordinates) so that the number of memory operations between a load and its use is
    // known at compile time: the wait in front of a step is then "all but the younger stages", not "everything".
    uint4 pre_pix[DEPTH]; uint32_t pre_rec[DEPTH], pre_ring[DEPTH];
    auto fetch = [&](int d, int s) {
        int xn = min(max(s - 2 * row, 0), mb_w - 1);
        if (CHAIN) {
            wait_recon(s - 2 * row);            // (steps outside the row load a clamped position whose data is never used: nothing to wait for)
            const JM_GLOBAL uint64_t *q = (const JM_GLOBAL uint64_t *)(pix_base + xn * 16);
            const uint64_t a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pre_pix[d] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
        } else
        pre_pix[d] = gload<uint4>(pix_base + xn * 16);
        pre_rec[d] = gload<uint32_t>(rec_base + (size_t)xn * sizeof(DbRec));
        if (band > 0) pre_ring[d] = __hip_atomic_load((const JM_GLOBAL uint32_t *)(ring_base + xn * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // ring rows of macroblock xm of this band's last row -> surface (write-through)
    auto give = [&](const uint8_t *ring, int xm, int lane0) {
        int k = l - lane0;
        if (k < 0 || k >= ring_lanes) return;
        uint32_t v = *(const uint32_t *)(ring + (k >> 2) * 16 + (k & 3) * 4);
        __hip_atomic_store((JM_GLOBAL uint32_t *)(plane + (size_t)((row + 1) * rows_per_mb - ring_rows + (k >> 2)) * pitch + xm * 16) + (k & 3), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto step = [&](int s, uint4 own, uint32_t rdw, uint32_t ring) {
        const int x = s - 2 * row;
        if (active && x >= 0 && x < mb_w) {
            if (is_chroma) {
                if (takes_ring && l < ring_lanes) *(uint32_t *)(lds.chroma_ring(0, x & 3) + (l >> 2) * 16 + (l & 3) * 4) = ring;
                chroma_mb<CHAIN>(cx, lds, x, row, lrow, l, group, own, rdw);
                if (gives_ring) {
                    if (x > 0) give(lds.chroma_ring(lrow, (x - 1) & 3), x - 1, 0);
                    if (x == mb_w - 1) give(lds.chroma_ring(lrow, x & 3), x, 8);
                }
            } else {
                if (takes_ring) *(uint32_t *)(lds.luma_ring(0, x & 3) + (l >> 2) * 16 + (l & 3) * 4) = ring;
                luma_mb<CHAIN>(cx, lds, x, row, lrow, l, group, own, rdw);
                if (gives_ring) {
                    if (x > 0) give(lds.luma_ring(lrow, (x - 1) & 3), x - 1, 0);
                    if (x == mb_w - 1) give(lds.luma_ring(lrow, x & 3), x, 0);
                }
            }
        }
        // publish: the wave that holds the band's last row wrote the ring rows itself, so waiting for ITS stores is enough
        if (gives_ring && pub > 0 && ((s + 1 - s_begin) % pub == 0 || s == s_end)) {        // pub <= 0: debug option "debug_stall" -- the counter never advances
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (l == 0) __hip_atomic_store(&prog[band], s == s_end ? 0x7fffffff : s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (CHAIN) {
            // `fin`: steps whose final samples are in memory, published two steps late.  Every wave has issued at least four memory
            // operations since its stores of step s - 2 (the prefetches of two steps), and a wave's operations retire in order, so "at most four
            // outstanding" means those stores have been acknowledged -- without draining the prefetches that were just issued.
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (threadIdx.x == 0 && s - 1 > s_begin) __hip_atomic_store(cpic + kChainFin + (is_chroma ? 32 : 0) + band, s - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) pre_ring[d] = 0;
    wait_above(s_begin + DEPTH);
#pragma unroll
    for (int d = 0; d < DEPTH; d++) fetch(d, s_begin + d);
    // stage j of the unrolled body always lives in the same registers: no copies of registers that still wait for their load
    for (int s = s_begin; s <= s_end; s += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            if (s + j > s_end) break;
            uint4 own = pre_pix[j]; uint32_t rdw = pre_rec[j], ring = pre_ring[j];
            asm volatile("" : "+v"(own.x), "+v"(own.y), "+v"(own.z), "+v"(own.w), "+v"(rdw), "+v"(ring));
            wait_above(s + j + DEPTH);                // the ring rows of step s + j + DEPTH are final once the band above completed the step before it
            fetch(j, s + j + DEPTH);
            step(s + j, own, rdw, ring);
        }
    }
    if (CHAIN) {                                    // everything this band stores is in memory
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_store(cpic + kChainFin + (is_chroma ? 32 : 0) + band, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace jmamd
