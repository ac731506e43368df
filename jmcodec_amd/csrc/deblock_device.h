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
// ---- the luma edge filter on PACKED 16-bit pairs (round 3) ----
// One 32-bit register holds the sample i of the p side in its low half and the sample i of the q side in its high half: A = (p0 | q0 << 16),
// B = (p1 | q1), C = (p2 | q2), D = (p3 | q3).  The two sides of 8.7.2.3 / 8.7.2.4 are mirror images, so every side-wise quantity (|p1 - p0| and
// |q1 - q0|, ap and aq, p1' and q1', the strong filter's three outputs per side) is ONE v_pk_* instruction instead of two scalar ones, conditions
// become masks ((x - limit) >> 15 per half) and selects become bitwise blends: ~60 vector instructions for the normal filter where the scalar
// form needed ~100, and no int <-> bool conversions.  Same arithmetic, value for value (tests: every deblocking case against the oracle).
typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2 as_s2(uint32_t v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ uint32_t as_u(s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s2 splat(int v) { return as_s2((uint32_t)v | ((uint32_t)v << 16)); }      // 0 <= v < 65536
__device__ __forceinline__ s2 swp(s2 v) { const uint32_t u = as_u(v); return as_s2(__builtin_amdgcn_alignbit(u, u, 16)); }
__device__ __forceinline__ s2 pabs(s2 v) { return __builtin_elementwise_max(v, -v); }
__device__ __forceinline__ s2 pmin(s2 a, s2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s2 pmax(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s2 blend(uint32_t m, s2 a, s2 b) { return as_s2((as_u(a) & m) | (as_u(b) & ~m)); }      // m: all ones / all zeros per half
// v_perm_b32: byte k of the result = byte sel[k] of {hi, lo} (0..3 = lo, 4..7 = hi, 0x0c = zero)
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// A, B, C are updated in place (D = p3 | q3 is only read); bsw = bS | tC0 << 3
__device__ __forceinline__ void flt_luma(s2 &A, s2 &B, s2 &C, const s2 D, int bsw, int alpha, int beta) {
    const int bS = bsw & 7, tc0 = bsw >> 3;
    const s2 As = swp(A), Bs = swp(B);
    const s2 beta2 = splat(beta);
    const s2 d10 = pabs(B - A), dpq = pabs(As - A);                       // (|p1 - p0| , |q1 - q0|), |p0 - q0| in both halves
    const s2 m10 = (d10 - beta2) >> 15, mpq = (dpq - splat(alpha)) >> 15;
    const uint32_t on = as_u(m10) & as_u(swp(m10)) & as_u(mpq) & (bS ? 0xffffffffu : 0u);      // filterSamplesFlag, the same in both halves
    if (!__builtin_amdgcn_ballot_w64(on != 0)) return;
    const s2 m20 = (pabs(C - A) - beta2) >> 15;                           // (ap , aq) as masks
    const int tc = tc0 + (int)(as_u(m20) & 1u) + (int)(as_u(m20) >> 31);
    const s2 t = ((As - A) << 2) + (B - Bs) + splat(4);                    // low half: ((q0 - p0) << 2) + (p1 - q1) + 4
    const int dl = clip3(-tc, tc, (int)(short)(as_u(t) & 0xffffu) >> 3);
    const s2 dd = as_s2(((uint32_t)dl & 0xffffu) | ((uint32_t)(-dl) << 16));       // (+delta , -delta)
    const s2 nA = pmin(pmax(A + dd, splat(0)), splat(255));
    const s2 avg = as_s2((as_u(A + As + splat(1)) >> 1) & 0x7fff7fffu);     // (p0 + q0 + 1) >> 1 in both halves
    const s2 tcs = splat(tc0);
    const s2 tt = pmin(pmax((C + avg - (B << 1)) >> 1, -tcs), tcs);
    const s2 nB = B + as_s2(as_u(tt) & as_u(m20));                        // p1' only with ap, q1' only with aq
    const uint32_t nrm = on & (bS < 4 ? 0xffffffffu : 0u);
    s2 rA = blend(nrm, nA, A), rB = blend(nrm, nB, B), rC = C;
    const uint32_t st = on & (bS >= 4 ? 0xffffffffu : 0u);
    if (__builtin_amdgcn_ballot_w64(st != 0)) {
        const s2 msg = (dpq - splat((alpha >> 2) + 2)) >> 15;
        const uint32_t sm = st & as_u(m20) & as_u(msg);                    // the strong filter, per side
        const s2 S0 = (C + ((B + A + As) << 1) + Bs + splat(4)) >> 3;
        const s2 S1 = (C + B + A + As + splat(2)) >> 2;
        const s2 S2 = ((D << 1) + C + (C << 1) + B + A + As + splat(4)) >> 3;
        const s2 W0 = ((B << 1) + A + Bs + splat(2)) >> 2;
        rA = blend(sm, S0, blend(st, W0, rA)); rB = blend(sm, S1, rB); rC = blend(sm, S2, rC);
    }
    A = rA; B = rB; C = rC;
}
// chroma: p1 p0 q0 q1 by reference
__device__ __forceinline__ void flt_chroma(int p1, int &p0, int &q0, int q1, int bsw, int alpha, int beta) {
    const int bS = bsw & 7, tc = (bsw >> 3) + 1;
    const bool on = ((int)(bS != 0) & (int)(adiff(p0, q0) < alpha) & (int)(adiff(p1, p0) < beta) & (int)(adiff(q1, q0) < beta)) != 0;
    const int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
    const int n_p0 = clip1(p0 + delta), n_q0 = clip1(q0 - delta), w_p0 = (2 * p1 + p0 + q1 + 2) >> 2, w_q0 = (2 * q1 + q0 + p1 + 2) >> 2;
    p0 = sel(on, sel(bS < 4, n_p0, w_p0), p0); q0 = sel(on, sel(bS < 4, n_q0, w_q0), q0);
}


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
// PHASE: 0 = the whole macroblock; 1 = its vertical edges only (kRowLag 1: every row of the band does this part of its step, then a barrier);
// 2 = horizontal edges + the final store
template <bool WT, int PHASE = 0>
__device__ __forceinline__ void luma_mb(const DbCtx &pp, const Lds &lds, int x, int row, int lrow, int l, int group, uint4 own, uint32_t recdw) {
    uint8_t *tc = lds.luma_tile(lrow, x & 1), *tp = lds.luma_tile(lrow, (x - 1) & 1);
    uint8_t *ring_up = row > 0 ? lds.luma_ring(lrow - 1, x & 3) : nullptr;
    uint8_t *ring_up_l = row > 0 ? lds.luma_ring(lrow - 1, (x - 1) & 3) : nullptr;
    uint8_t *ring_dn = lds.luma_ring(lrow, x & 3), *ring_dn_l = lds.luma_ring(lrow, (x - 1) & 3);
    uint8_t *rec = lds.rec(group);
    if (PHASE != 2 && l < 12) ((uint32_t *)rec)[l] = recdw;   // 48-byte luma half of the DbRec
    // this lane's four vertical-edge and four horizontal-edge strengths (bS | tC0 << 3), and the class parameters
    int vb[4], hb[4], ab[6];
#pragma unroll
    for (int e = 0; e < 4; e++) { if (PHASE != 2) vb[e] = rec[e * 4 + (l >> 2)]; if (PHASE != 1) hb[e] = rec[16 + e * 4 + (l >> 2)]; }
#pragma unroll
    for (int i = 0; i < 6; i++) ab[i] = rec[32 + i];
    if (PHASE != 2) {
    // ---- vertical edges: lane = pixel row l ----
    uint32_t left = x > 0 ? *(const uint32_t *)(tp + l * 16 + 12) : 0;
    // The row's 20 samples x[-4 .. 15] arrive as five dwords.  Edge e pairs x[4e - 1 - i] with x[4e + i]: one v_perm_b32 per pair -- the p side of
    // edge 0 from the left neighbour's dword, that of a later edge from the HIGH halves the previous edge left behind (its q side, filtered).
    uint32_t left_after, row_after[4];
    {
        const uint32_t w[5] = {left, own.x, own.y, own.z, own.w};
        s2 A = as_s2(perm(w[1], w[0], 0x0c040c03)), B = as_s2(perm(w[1], w[0], 0x0c050c02)), C = as_s2(perm(w[1], w[0], 0x0c060c01)),
            D = as_s2(perm(w[1], w[0], 0x0c070c00));
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (e) {
                const s2 pA = A, pB = B, pC = C, pD = D;
                A = as_s2(perm(w[e + 1], as_u(pD), 0x0c040c02)); B = as_s2(perm(w[e + 1], as_u(pC), 0x0c050c02));
                C = as_s2(perm(w[e + 1], as_u(pB), 0x0c060c02)); D = as_s2(perm(w[e + 1], as_u(pA), 0x0c070c02));
            }
            const int c = e ? 1 : 0;
            flt_luma(A, B, C, D, vb[e], ab[2 * c], ab[2 * c + 1]);
            // x[4e - 4 .. 4e - 1] are final for this pass: the low halves of D, C, B, A
            const uint32_t o = perm(as_u(C), as_u(D), 0x0c0c0400) | perm(as_u(A), as_u(B), 0x04000c0c);
            if (e == 0) left_after = o; else row_after[e - 1] = o;
        }
        row_after[3] = perm(as_u(B), as_u(A), 0x0c0c0602) | perm(as_u(D), as_u(C), 0x06020c0c);      // x[12 .. 15]: the high halves of edge 3
    }
    if (x > 0) *(uint32_t *)(tp + l * 16 + 12) = left_after;
    *(uint4 *)(tc + l * 16) = make_uint4(row_after[0], row_after[1], row_after[2], row_after[3]);
    // the left neighbour's bottom rows (its ring slot) get our edge-0 result for columns 12..15
    if (x > 0 && l >= 12) *(uint32_t *)(ring_dn_l + (l - 12) * 16 + 12) = left_after;
    }
    if (PHASE == 1) return;
    // ---- horizontal edges: lane = pixel column l ----
    {
        uint32_t c[20];
        c[0] = c[1] = c[2] = c[3] = 0;
        if (ring_up) { c[0] = ring_up[l]; c[1] = ring_up[16 + l]; c[2] = ring_up[32 + l]; c[3] = ring_up[48 + l]; }      // (one region, not four)
#pragma unroll
        for (int j = 0; j < 16; j++) c[4 + j] = tc[j * 16 + l];
        s2 A = as_s2(c[3] | (c[4] << 16)), B = as_s2(c[2] | (c[5] << 16)), C = as_s2(c[1] | (c[6] << 16)), D = as_s2(c[0] | (c[7] << 16));
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (e) {
                const s2 pA = A, pB = B, pC = C, pD = D;
                A = as_s2(perm(c[4 * e + 4], as_u(pD), 0x0c040c02)); B = as_s2(perm(c[4 * e + 5], as_u(pC), 0x0c040c02));
                C = as_s2(perm(c[4 * e + 6], as_u(pB), 0x0c040c02)); D = as_s2(perm(c[4 * e + 7], as_u(pA), 0x0c040c02));
            }
            const int k = e ? 1 : 2;
            flt_luma(A, B, C, D, hb[e], ab[2 * k], ab[2 * k + 1]);
            // rows 4e - 4 .. 4e - 1 of the column are final: the low halves of D, C, B, A (row 4e - 4 = p3 is never changed by this edge)
            if (e == 0) { if (ring_up) { ring_up[1 * 16 + l] = (uint8_t)as_u(C); ring_up[2 * 16 + l] = (uint8_t)as_u(B);
                ring_up[3 * 16 + l] = (uint8_t)as_u(A); } }
            else { const int r = 4 * e - 4; tc[r * 16 + l] = (uint8_t)as_u(D); tc[(r + 1) * 16 + l] = (uint8_t)as_u(C);
                tc[(r + 2) * 16 + l] = (uint8_t)as_u(B); tc[(r + 3) * 16 + l] = (uint8_t)as_u(A); }
        }
        // rows 12 .. 15: the high halves of edge 3 -- into the tile and, as the rows the macroblock below starts from, into this row's ring slot
        const uint8_t r12 = (uint8_t)(as_u(A) >> 16), r13 = (uint8_t)(as_u(B) >> 16), r14 = (uint8_t)(as_u(C) >> 16), r15 = (uint8_t)(as_u(D) >> 16);
        tc[12 * 16 + l] = r12; tc[13 * 16 + l] = r13; tc[14 * 16 + l] = r14; tc[15 * 16 + l] = r15;
        ring_dn[0 * 16 + l] = r12; ring_dn[1 * 16 + l] = r13; ring_dn[2 * 16 + l] = r14; ring_dn[3 * 16 + l] = r15;
    }
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
template <bool WT, int PHASE = 0>
__device__ __forceinline__ void chroma_mb(const DbCtx &pp, const Lds &lds, int x, int row, int lrow, int l, int group, uint4 own, uint32_t recdw) {
    uint8_t *tc = lds.chroma_tile(lrow, x & 1), *tp = lds.chroma_tile(lrow, (x - 1) & 1);
    uint8_t *ring_up = row > 0 ? lds.chroma_ring(lrow - 1, x & 3) : nullptr;
    uint8_t *ring_up_l = row > 0 ? lds.chroma_ring(lrow - 1, (x - 1) & 3) : nullptr;
    uint8_t *ring_dn = lds.chroma_ring(lrow, x & 3), *ring_dn_l = lds.chroma_ring(lrow, (x - 1) & 3);
    uint8_t *rec = lds.rec(group);
    if (PHASE != 2 && l < 12) ((uint32_t *)rec)[l] = recdw;   // 48-byte chroma half of the DbRec
    // ---- vertical edges (chroma columns 0 and 4 <-> luma edges 0 and 2): lane = (plane, chroma row) ----
    if (PHASE != 2) {
        const int plane = l >> 3, r = l & 7;
        // `own` of lane l is row (l & 7) of the macroblock (both lane halves prefetch the same 16 bytes)
        const uint32_t left = x > 0 ? *(const uint32_t *)(tp + r * 16 + 12) : 0;
        if (plane == 0) *(uint4 *)(tc + r * 16) = own;
        // Cb and Cr bytes alternate: sample k of this lane's plane in a dword is byte 2k + plane -- taken with a shift by 8 * plane.  (Indexing an
        // unpacked register array with the run-time `plane` made the compiler build every access as a chain of twenty compares and selects:
        // about a thousand instructions per step, three quarters of this workgroup's code.)
        const int s0 = plane * 8, s1 = s0 + 16;
        const int vb[2] = { rec[plane * 16 + (r >> 1)], rec[plane * 16 + 4 + (r >> 1)] };
        const int ab[4] = { rec[32 + plane * 6], rec[32 + plane * 6 + 1], rec[32 + plane * 6 + 2], rec[32 + plane * 6 + 3] };
        {   // edge 0: p1 p0 | q0 q1 = left dword samples 0, 1 | own.x samples 0, 1
            int p0 = (int)(left >> s1) & 255, q0 = (int)(own.x >> s0) & 255;
            flt_chroma((int)(left >> s0) & 255, p0, q0, (int)(own.x >> s1) & 255, vb[0], ab[0], ab[1]);
            if (vb[0] & 7) { tp[r * 16 + 14 + plane] = (uint8_t)p0; tc[r * 16 + plane] = (uint8_t)q0; }
        }
        {   // edge 2 (chroma column 4): own.y samples 0, 1 | own.z samples 0, 1
            int p0 = (int)(own.y >> s1) & 255, q0 = (int)(own.z >> s0) & 255;
            flt_chroma((int)(own.y >> s0) & 255, p0, q0, (int)(own.z >> s1) & 255, vb[1], ab[2], ab[3]);
            if (vb[1] & 7) { tc[r * 16 + 6 + plane] = (uint8_t)p0; tc[r * 16 + 8 + plane] = (uint8_t)q0; }
        }
    }
    // left neighbour's bottom rows: columns 12..15 (bytes) of rows 6, 7 after our edge 0
    if (PHASE != 2 && x > 0 && l < 2) *(uint32_t *)(ring_dn_l + l * 16 + 12) = *(const uint32_t *)(tp + (6 + l) * 16 + 12);
    if (PHASE == 1) return;
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
// resident at once (the first form of this kernel walked a whole plane with ONE 16-wave workgroup and needed 159 KB of LDS). A band runs the same steps s = x +
// 2 * row for its own rows; the only
// coupling is downwards: the last row of a band hands the bottom four sample rows of each macroblock (after its own filtering, i.e. the
// state clause 8.7 prescribes when the macroblock below starts) to the first row of the next band.  They travel through the picture
// surface itself (the band below overwrites them with the final values afterwards) and a per-band step counter in device memory:
//     band b publishes "steps completed" every `pub` steps (release); band b+1 polls it (acquire) only when the value it last saw does not
//     cover the step it is about to prefetch for, so in steady state it runs a few steps behind without waiting.
// Workgroups of a lower band have the higher block index, so whatever a workgroup waits for has been dispatched before it.
constexpr int kBandRows = 16;
// kRowLag (chain_common.h): steps between a macroblock row and the row below it -- 1 since round 4 (a step = V phase | barrier | H phase)
constexpr int kDeblockSmemMain = kBandRows * Lds::kRecStride + (kBandRows + 1) * (Lds::kLT + Lds::kLR);
// Per wave and prefetch stage, the inputs of one step as the LDS-DMA loads deliver them (lane-linear): 64 x 16 B of samples, 64 x 4 B of DbRec
// dwords, 64 x 4 B of ring-row dwords from the band above.
constexpr int kStageBytes = 1024 + 256 + 256, kDeblockMaxDepth = 4;
constexpr int kDeblockSmemBytes = kDeblockSmemMain + (kBandRows / 4) * kDeblockMaxDepth * kStageBytes;

// ---- LDS-DMA loads issued from inline assembly (round 3) ----
// What bounded a step of this kernel was not its arithmetic (switching the filters off changed nothing) but the memory waits hipcc places: with the
// step's conditional stores between a prefetch and its use it cannot count the operations in flight, so every step began with s_waitcnt vmcnt(0 / 1)
// -- a full round trip to memory for the stores issued a moment ago -- and the per-step barrier handed the slowest wave's wait to all four
// (profiles/README.md: 2.7 us per step; the same pipeline with counted waits: 0.64 us in scratch/exp/glds_probe.hip).  A load that writes LDS directly
// has no register destination the compiler could touch early, so it can be issued from inline assembly, invisible to hipcc's bookkeeping, and
// retired by a COUNTED s_waitcnt of our own: the operations of a wave retire in order, and between a stage's loads and their use every wave issues
// at least the loads of the younger stages, whatever it stores in between.  M0 (the LDS destination base) is saved and restored around the load
// (cdna_hip_programming.md, inline assembly, LDS-DMA recipe).  The destination is wave-uniform base + lane x size.
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)p; }
template <bool COHERENT> __device__ __forceinline__ void glds16(const gbyte *g, uint32_t lds_byte_addr) {
    unsigned keep;
    if (COHERENT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(lds_byte_addr) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g),
        "s"(lds_byte_addr) : "memory");
}
template <bool COHERENT> __device__ __forceinline__ void glds4(const gbyte *g, uint32_t lds_byte_addr) {
    unsigned keep;
    if (COHERENT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(lds_byte_addr) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g),
        "s"(lds_byte_addr) : "memory");
}
// One 256-thread workgroup deblocks band `band` of one plane of picture pp.  prog = the picture's band step counters of the ring-row hand-over
// (kDeblockProgressStride ints).  CHAIN (k_chain): `cpic` = the picture's block of the chain buffer; the unfiltered samples come from
// reconstruction waves of the same launch (wait for their bits, read with cache-bypassing loads), the final samples are written through
// and the band publishes in cpic[kChainFin ..] how many of its steps are final in memory.
// AFTER_INTRA (k_chain_i only): the picture's unfiltered samples come from an intra wavefront of the same launch (see wait_recon below)
template <int DEPTH, bool CHAIN, bool AFTER_INTRA = false>
__device__ __forceinline__ void deblock_band_body(const PicParams &pp, int band, bool is_chroma, int *prog_pic, int pub, uint8_t *smem, int *cpic,
    int *err_word) {
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
    gbyte *plane = (gbyte *)(cur_plane(pp) + (is_chroma ? pp.chroma_offset : 0));
    const DbCtx cx{plane, pitch, mb_w, mb_h};
    const int rows_per_mb = is_chroma ? 8 : 16, ring_rows = is_chroma ? 2 : 4;
    const int my_row = is_chroma ? (l & 7) : l;
    const int rec_dw = (is_chroma ? 12 : 0) + (l < 12 ? l : 0);
    const bool takes_ring = band > 0 && group == 0;                        // first row of a lower band: ring rows come from the band above
    const bool gives_ring = active && group == rows - 1 && row < mb_h - 1;  // last row of a band that has a band below
    // (kRowLag 1: one more step, the H + store of the last macroblock of the band's last row)
    const int s_begin = kRowLag * row0, s_end = mb_w - 1 + kRowLag * (row0 + rows - 1) + (kRowLag == 1 ? 1 : 0);
    int known = 0;                                                          // steps the band above is known to have completed
    // launch-wide: a wait of this launch gave up (chain_common.h)
    int *abort_word = CHAIN ? cpic - (size_t)pp.chain_idx * kChainStride + (size_t)kChainMaxPics * kChainStride : nullptr;
    // Hand-over protocol without cache maintenance: the ring rows and the counter are written and read with agent-scope relaxed atomics
    // (write-through stores / loads that bypass the non-coherent cache levels), ordered by a plain s_waitcnt on the writer's side and by the
    // data dependency on the counter on the reader's side.  Acquire / release at agent scope would write back and invalidate the whole
    // L2 of the XCD on every poll (measured: 2.8 ms per launch with one release per step).
    auto wait_above = [&](int need) {
        if (band == 0 || threadIdx.x >= 64) return;                          // wave 0 holds group 0
        if (known >= need) return;
        int spins = 0; WaitClock t0;
        while ((known = __hip_atomic_load(&prog[band - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need && ++spins < kSpinLimit &&
               !(CHAIN && (wait_expired(spins, t0, abort_word) || ((spins & 255) == 0 && ld_coh(abort_word))))) __builtin_amdgcn_s_sleep(8);
        if (CHAIN && l == 0) note_gaps(t0, err_word ? err_word - pp.chain_idx + kChainMaxPics : nullptr);
        if (known < need) {                                                  // the band above never got there: damaged, and SAID so
            if (l == 0) { report_wait_timeout(err_word, CHAIN_ERR_RING_TIMEOUT);
                if (CHAIN) {
                    if (!ld_coh(abort_word)) { record_first_giveup(abort_word, CHAIN_ERR_RING_TIMEOUT, pp.chain_idx, band << 16 | (is_chroma ? 1 : 0), need,
                        known, 0);
                        record_giveup_evidence(abort_word, &prog[band - 1], spins, t0); }
                    st_coh(abort_word, 1);
                } }
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
        const int tag = pp.chain_idx << 16 | (row & 0xffff);
        if (after_intra) { const int need = x + 2 * row + 4; ok = wait_counter(ifin0, ifin_known0, want, need, abort_word, tag) &&
            wait_counter(ifin1, ifin_known1, want, need, abort_word, tag); }
        else ok = wait_row_bit(bits_row, recon_known, want, x, abort_word, tag, err_word ? err_word - pp.chain_idx + kChainMaxPics : nullptr);
        if (!ok && (threadIdx.x & 63) == 0) { report_wait_timeout(err_word, after_intra ? CHAIN_ERR_IFIN_TIMEOUT : CHAIN_ERR_BITS_TIMEOUT);
            st_coh(abort_word, 1); }
    };
    const int ring_lanes = ring_rows * 4;                                   // one dword per lane: ring row l >> 2, dword l & 3
    const int ring_lane = l < ring_lanes ? l : 0;
    const gbyte *pix_base = plane + (size_t)(row * rows_per_mb + my_row) * pitch;
    const gbyte *rec_base = recs + (size_t)row * mb_w * sizeof(DbRec) + rec_dw * 4;
    const gbyte *ring_base = plane + (size_t)(takes_ring ? row * rows_per_mb - ring_rows + (ring_lane >> 2) : 0) * pitch + (ring_lane & 3) * 4;
    // The loads of a stage are unconditional (clamped coordinates; band 0 "loads" ring dwords it never uses): every wave issues exactly three
    // LDS-DMA loads per step, which is what makes the counted waits below exact enough.
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wl = (int)(threadIdx.x & 63);
    uint8_t *stage = smem + kDeblockSmemMain + wave * (kDeblockMaxDepth * kStageBytes);
    const uint32_t stage_lds = lds_addr_of(stage);
    auto fetch = [&](int d, int s) {
        int xn = min(max(s - kRowLag * row, 0), mb_w - 1);
        if (CHAIN) wait_recon(s - kRowLag * row);     // (steps outside the row load a clamped position whose data is never used: nothing to wait for)
        const uint32_t slot = stage_lds + (uint32_t)d * kStageBytes;
        glds16<CHAIN>(pix_base + xn * 16, slot);                                        // CHAIN: reconstructed in this launch -> coherent load
        glds4<false>(rec_base + (size_t)xn * sizeof(DbRec), slot + 1024);
        glds4<true>(band > 0 ? ring_base + xn * 16 : rec_base, slot + 1280);
    };
    // ring rows of macroblock xm of this band's last row -> surface (write-through)
    auto give = [&](const uint8_t *ring, int xm, int lane0) {
        int k = l - lane0;
        if (k < 0 || k >= ring_lanes) return;
        uint32_t v = *(const uint32_t *)(ring + (k >> 2) * 16 + (k & 3) * 4);
        __hip_atomic_store((JM_GLOBAL uint32_t *)(plane + (size_t)((row + 1) * rows_per_mb - ring_rows + (k >> 2)) * pitch + xm * 16) + (k & 3), v,
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto step = [&](int s, uint4 own, uint32_t rdw, uint32_t ring) {
        if (kRowLag == 1) {
            // One barrier per step, and one row per step (chain_common.h kRowLag): the step is NOT "macroblock x" but "H + store of macroblock x - 1, then V
            // of macroblock x" of every row.  H(x - 1, row) needs V(x, row - 1) -- the left edge of the macroblock above right changes columns 13..15 of
            // the ring rows it starts from -- and that ran in the PREVIOUS step of the row above (row - 1 is one macroblock ahead), i.e. behind the
            // barrier that ended it; V(x, row) follows H(x - 1, row) in program order.  (The first version of the one-row schedule ran V | barrier | H
            // inside a step: two barriers, each paying the slowest of the four waves -- 3.0 us per step against 2.6.)
            const int xv = s - row, xh = xv - 1;
            if (active && xh >= 0 && xh < mb_w) {
                if (is_chroma) { chroma_mb<CHAIN, 2>(cx, lds, xh, row, lrow, l, group, own, rdw);
                    if (gives_ring && xh == mb_w - 1) give(lds.chroma_ring(lrow, xh & 3), xh, 8); }
                else { luma_mb<CHAIN, 2>(cx, lds, xh, row, lrow, l, group, own, rdw);
                    if (gives_ring && xh == mb_w - 1) give(lds.luma_ring(lrow, xh & 3), xh, 0); }
            }
            if (active && xv >= 0 && xv < mb_w) {
                if (is_chroma) {
                    if (takes_ring && l < ring_lanes) *(uint32_t *)(lds.chroma_ring(0, xv & 3) + (l >> 2) * 16 + (l & 3) * 4) = ring;
                    chroma_mb<CHAIN, 1>(cx, lds, xv, row, lrow, l, group, own, rdw);
                    if (gives_ring && xv > 0) give(lds.chroma_ring(lrow, (xv - 1) & 3), xv - 1, 0);
                } else {
                    if (takes_ring) *(uint32_t *)(lds.luma_ring(0, xv & 3) + (l >> 2) * 16 + (l & 3) * 4) = ring;
                    luma_mb<CHAIN, 1>(cx, lds, xv, row, lrow, l, group, own, rdw);
                    if (gives_ring && xv > 0) give(lds.luma_ring(lrow, (xv - 1) & 3), xv - 1, 0);
                }
            }
        } else {
            const int x = s - kRowLag * row;
            if (active && x >= 0 && x < mb_w) {
                if (is_chroma) {
                    if (takes_ring && l < ring_lanes) *(uint32_t *)(lds.chroma_ring(0, x & 3) + (l >> 2) * 16 + (l & 3) * 4) = ring;
                    chroma_mb<CHAIN, 0>(cx, lds, x, row, lrow, l, group, own, rdw);
                    if (gives_ring) {
                        if (x > 0) give(lds.chroma_ring(lrow, (x - 1) & 3), x - 1, 0);
                        if (x == mb_w - 1) give(lds.chroma_ring(lrow, x & 3), x, 8);
                    }
                } else {
                    if (takes_ring) *(uint32_t *)(lds.luma_ring(0, x & 3) + (l >> 2) * 16 + (l & 3) * 4) = ring;
                    luma_mb<CHAIN, 0>(cx, lds, x, row, lrow, l, group, own, rdw);
                    if (gives_ring) {
                        if (x > 0) give(lds.luma_ring(lrow, (x - 1) & 3), x - 1, 0);
                        if (x == mb_w - 1) give(lds.luma_ring(lrow, x & 3), x, 0);
                    }
                }
            }
        }
        // publish: the wave that holds the band's last row wrote the ring rows itself, so waiting for ITS stores is enough -- and not by draining
        // the wave: it has issued the three loads of each of the last kPubLag steps since the ring stores of step s - kPubLag (the loads of a step go
        // out before its stores), operations retire in order, so "at most 3 * kPubLag outstanding" proves those stores acknowledged.  The band below
        // sees the counter kPubLag steps late; it runs some thirty steps behind anyway.
        constexpr int kPubLag = 3;
        // pub <= 0: debug option "debug_stall" -- the counter never advances
        if (gives_ring && pub > 0 && s - kPubLag >= s_begin && ((s - kPubLag + 1 - s_begin) % pub == 0)) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * kPubLag) : "memory");
            if (l == 0) __hip_atomic_store(&prog[band], s - kPubLag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (gives_ring && pub > 0 && s == s_end) {                                           // the band's last step: everything, then "done"
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (l == 0) __hip_atomic_store(&prog[band], 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (CHAIN) {
            // `fin`: how many steps have their final samples in memory, published two steps late by the same counting argument: every wave has issued
            // the six loads of the steps s - 1 and s since its stores of step s - 2.
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (threadIdx.x == 0 && s - 1 > s_begin) __hip_atomic_store(cpic + kChainFin + (is_chroma ? 32 : 0) + band, s - 1, __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
        } else
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    static_assert(DEPTH >= 2 && DEPTH <= kDeblockMaxDepth, "prefetch depth");
    // the ring rows a step starts from are final once the band above has completed the step before it (kRowLag 2) / that very step (kRowLag 1: its last
    // row runs V(x + 1) in the step in which this band's first row runs H(x))
    constexpr int kAbove = 2 - kRowLag;
    wait_above(s_begin + DEPTH + kAbove);
#pragma unroll
    for (int d = 0; d < DEPTH; d++) fetch(d, s_begin + d);
    for (int s = s_begin; s <= s_end; s += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            if (s + j > s_end) break;
            // the loads of this stage were issued DEPTH steps ago; since then the wave has issued at least the 3 * (DEPTH - 1) loads of the younger
            // stages (plus whatever it stored): in-order retirement makes "at most that many outstanding" sufficient
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (DEPTH - 1)) : "memory");
            const uint8_t *sl = stage + j * kStageBytes;
            const uint4 own = *(const uint4 *)(sl + wl * 16);
            const uint32_t rdw = *(const uint32_t *)(sl + 1024 + wl * 4), ring = *(const uint32_t *)(sl + 1280 + wl * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot is read before its next load is even issued
            wait_above(s + j + DEPTH + kAbove);      // the ring rows of step s + j + DEPTH: see above
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
