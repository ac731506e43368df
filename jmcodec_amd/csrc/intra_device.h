// jmcodec_amd/csrc/intra_device.h -- device code of the LDS-resident intra prediction wavefront (H.264 8.3), shared by k_intra_band
// (intra_lds.hip: one launch per stage) and k_chain (chain.hip: the I picture of an IDR period as the first picture of a chain launch).
// See intra_lds.hip for the algorithm; part of the replacement for cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:33-41).
#pragma once
#include <hip/hip_runtime.h>
#include "jobs.h"
#include "kernels.h"
#include "kernel_common.h"
#include "chain_common.h"
#include "intra8_packed.h"

namespace jmamd {

// The luma work tile of a macroblock: 17 rows of kTS bytes.  Row 0 = the sample row above the macroblock, rows 1..16 = its own.  In a row the
// neighbour column (left sample / corner) sits at byte kTO, the 16 samples at kTO + 1 .. kTO + 16 -- i.e. at dword-aligned offsets 4 .. 19, so the four
// samples above a 4x4 block, the four above right and a whole reconstructed row are aligned dword reads (round 4) -- and up to 8 samples above right
// (Intra8x8) behind them.
constexpr int kTS = 28, kTO = 3;
constexpr int kI4SelBase = 3904;     // Intra4x4 selector table (i4_sel_entry): 9 modes x 16 pixels x 8 bytes
constexpr int kI8SelBase = 3904 + 1152;  // Intra8x8 selector table (intra8_packed.h pk::I8Sel): 9 modes x 16 lanes x 64 bytes
constexpr int kTileBase = kI8SelBase + 9 * 16 * 64;   // tables and record staging come first

// (c, kind) of Intra4x4 mode `mode` for pixel (x, y); kind 0 copy P[c], 1 two-tap (P[c]+P[c+1]+1)>>1,
// 2 three-tap (P[c-1]+2P[c]+P[c+1]+2)>>2, 3 DC.  Edge path index: 0 L3' 1 L3 2 L2 3 L1 4 L0 5 TL 6..13 T0..T7 14 T7'
__device__ __forceinline__ int i4_table_entry(int mode, int x, int y) {
    int c = 0, kind = 0;
    switch (mode) {
    case 0: c = 6 + x; kind = 0; break;
    case 1: c = 4 - y; kind = 0; break;
    case 2: c = 0; kind = 3; break;
    case 3: c = 7 + x + y; kind = 2; break;
    case 4: c = 5 + x - y; kind = 2; break;
    case 5: { int z = 2 * x - y, i = x - (y >> 1);
        if (z >= 0) { c = 5 + i; kind = (z & 1) ? 2 : 1; }
        else if (z == -1) { c = 5; kind = 2; }
        else { c = 6 - y; kind = 2; }
        break; }
    case 6: { int z = 2 * y - x, i = y - (x >> 1);
        if (z >= 0) { if (z & 1) { c = 5 - i; kind = 2; } else { c = 4 - i; kind = 1; } }
        else if (z == -1) { c = 5; kind = 2; }
        else { c = 4 + x; kind = 2; }
        break; }
    case 7: { int i = x + (y >> 1); if (y & 1) { c = 7 + i; kind = 2; } else { c = 6 + i; kind = 1; } break; }
    default: { int z = x + 2 * y, i = y + (x >> 1);
        if (z > 5) { c = 1; kind = 0; }
        else if (z == 5) { c = 1; kind = 2; }
        else if (z & 1) { c = 3 - i; kind = 2; }
        else { c = 3 - i; kind = 1; }
        break; }
    }
    return c | (kind << 4);
}

// Round 4: the same table as byte selectors for v_perm_b32.  A lane holds the 15-entry edge path of its block in four registers P0..P3 (entry k = byte
// k); its three taps are the entries c - 1, c, c + 1 (entries 0 and 14 duplicate 1 and 13, which is the clamping of 8.3.1.2).  selA takes them out of
// P1:P0 (entries 0..7), selB out of P3:P2 (entries 8..15); a selector byte 0x0c yields zero, so the OR of the two v_perm results has tap t in byte t.
// Byte 3 of selA carries `kind` (it selects some byte of P0 into byte 3 of the result, which is never looked at).
__device__ __forceinline__ uint2 i4_sel_entry(int mode, int x, int y) {
    const int e = i4_table_entry(mode, x, y), c = e & 15, kind = e >> 4;
    uint32_t a = (uint32_t)kind << 24, b = 0x0c000000u;
    for (int t = 0; t < 3; t++) {
        const int k = kind == 3 ? 1 : c - 1 + t;                    // (DC reads no tap)
        a |= (uint32_t)(k < 8 ? k : 0x0c) << (8 * t);
        b |= (uint32_t)(k >= 8 ? k - 8 : 0x0c) << (8 * t);
    }
    return make_uint2(a, b);
}

// LDS image of one workgroup
struct ILds {
    uint8_t *base; int mb_h;
    // common
    __device__ uint8_t *i4tab() const { return base; }                                            // 144 B
    __device__ uint8_t *rec(int g) const { return base + 256 + g * 32; }                          // MbRec staging, 32 groups
    // (bytes 1280 .. 3903 held the Intra8x8 byte table and per-group edge buffers of rounds 1-4; the path lives in registers now)
    __device__ uint8_t *i8sel() const { return base + kI8SelBase; }                               // 9216 B: pk::I8Sel per (mode, lane)
    __device__ uint8_t *i4sel() const { return base + kI4SelBase; }                               // 1152 B: uint2 per (mode, pixel)
    // luma: tile[17][kTS] per group (476 -> 480), residual [16][16] int16 per group (512)
    __device__ uint8_t *ltile(int g) const { return base + kTileBase + g * 480; }
    __device__ short *lres(int g) const { return (short *)(base + kTileBase + 32 * 480 + g * 512); }
    __device__ uint8_t *lrcol(int row) const { return base + kTileBase + 32 * 992 + row * 16; }
    __device__ uint8_t *lring(int row, int slot) const { return base + kTileBase + 32 * 992 + mb_h * 16 + row * 64 + slot * 16; }
    // chroma: tile [8][16] interleaved per group (128), right column [8][2] per row, ring 4 x 16 B per row
    __device__ uint8_t *ctile(int g) const { return base + kTileBase + g * 128; }
    __device__ uint8_t *crcol(int row) const { return base + kTileBase + 32 * 128 + row * 16; }
    __device__ uint8_t *cring(int row, int slot) const { return base + kTileBase + 32 * 128 + mb_h * 16 + row * 64 + slot * 16; }
};

// What a macroblock step needs of the picture, read once per workgroup; the plane pointer is a global (address space 1) pointer so that the
// stores are global_store, not FLAT (see deblock_lds.hip).
typedef __attribute__((address_space(1))) uint8_t gbyte;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define JM_GLOBAL __attribute__((address_space(1)))
struct ICtx { gbyte *plane; int pitch; };
__device__ __forceinline__ void gstore4(gbyte *p, uint4 v) { v4u t = {v.x, v.y, v.z, v.w}; *(JM_GLOBAL v4u *)p = t; }
// WT: chain launches (chain_common.h)
template <bool WT> __device__ __forceinline__ void istore4(gbyte *p, uint4 v) { if (WT) st_wt16((void *)p, v); else gstore4(p, v); }
// cache-bypassing loads (chain launches): 8-byte relaxed agent-scope atomics
__device__ __forceinline__ uint4 gload4_coh(const gbyte *p) {
    const JM_GLOBAL uint64_t *q = (const JM_GLOBAL uint64_t *)p;
    const uint64_t a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
        b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}
__device__ __forceinline__ uint32_t gload1_coh(const gbyte *p) { return __hip_atomic_load((const JM_GLOBAL uint32_t *)p, __ATOMIC_RELAXED,
    __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint4 gload4(const gbyte *p) { v4u v = *(const JM_GLOBAL v4u *)p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint32_t gload1(const gbyte *p) { return *(const JM_GLOBAL uint32_t *)p; }

// sum over the 16 (or n) lanes of a group
__device__ __forceinline__ int group_sum16(int v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ int sum8(int v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); return v; }
__device__ __forceinline__ int sum4(int v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); return v; }

// ------------------------------------------------------------------------------------------
// luma, one macroblock, 16 lanes.  res0/res1: this lane's residual row (16 int16); right4/bottom: prefetched
// samples of an already reconstructed (non-intra) macroblock: columns 12..15 of row l, and the whole row 15.
// ------------------------------------------------------------------------------------------
// row = macroblock row in the picture (addresses in the surface), lrow = the row's index in the workgroup's LDS image (band-local + 1; index 0
// holds the bottom rows handed down by the band above)
template <bool WT>
__device__ __forceinline__ void intra_luma_mb(const ICtx &pp, const ILds &lds, int x, int row, int lrow, int l, int g,
                              uint32_t recdw, uint4 res0, uint4 res1, uint32_t right4, uint4 bottom) {
    uint8_t *rcol = lds.lrcol(lrow);
    uint8_t *ring_dn = lds.lring(lrow, x & 3);
    // MbRec through LDS (8 dwords); every lane needs kind / modes / flags / i4 modes
    uint32_t *rec = (uint32_t *)lds.rec(g);
    if (l < 8) rec[l] = recdw;
    uint32_t r0 = rec[0];
    int kind = r0 & 255, modes = (r0 >> 16) & 255, flags = r0 >> 24;
    if (kind != MB_I4 && kind != MB_I16) {
        // reconstructed earlier: publish its right column and bottom row for the neighbours
        rcol[l] = (uint8_t)(right4 >> 24);
        if (l < 4) ((uint32_t *)ring_dn)[l] = l == 0 ? bottom.x : (l == 1 ? bottom.y : (l == 2 ? bottom.z : bottom.w));
        return;
    }
    bool availA = flags & MBF_AVAIL_A, availB = flags & MBF_AVAIL_B, availC = flags & MBF_AVAIL_C, availD = flags & MBF_AVAIL_D;
    gbyte *dst = pp.plane + (size_t)(row * 16 + l) * pp.pitch + x * 16;
    const uint8_t *ring_up = row > 0 ? lds.lring(lrow - 1, x & 3) : ring_dn;           // only read when availB
    const uint8_t *ring_ur = row > 0 ? lds.lring(lrow - 1, (x + 1) & 3) : ring_dn;
    const uint8_t *ring_ul = row > 0 ? lds.lring(lrow - 1, (x - 1) & 3) : ring_dn;
    int rs[16];
    { uint32_t w[8] = {res0.x, res0.y, res0.z, res0.w, res1.x, res1.y, res1.z, res1.w};
#pragma unroll
      for (int i = 0; i < 16; i++) rs[i] = (int)(short)(w[i >> 1] >> ((i & 1) * 16)); }
    int left = rcol[l], corner = ring_ul[15];
    uint32_t o0, o1, o2, o3;                                      // this lane's reconstructed row
    if (kind == MB_I16) {
        int out[16];
        int mode = (modes >> 2) & 3;
        uint4 tv = *(const uint4 *)ring_up;
        int T[16];
        { uint32_t w[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
          for (int i = 0; i < 16; i++) T[i] = (w[i >> 2] >> ((i & 3) * 8)) & 255; }
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) out[i] = T[i];
        } else if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) out[i] = left;
        } else if (mode == 2) {
            int st = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) st += T[i];
            int sl = group_sum16(left);
            int dc = (availA && availB) ? (st + sl + 16) >> 5 : (availA ? (sl + 8) >> 4 : (availB ? (st + 8) >> 4 : 128));
#pragma unroll
            for (int i = 0; i < 16; i++) out[i] = dc;
        } else {
            int Hs = -8 * corner;
#pragma unroll
            for (int i = 0; i < 16; i++) Hs += (i - 7) * T[i];
            int Vs = group_sum16((l - 7) * left) - 8 * corner;
            int l15 = __shfl(left, (threadIdx.x & 48) | 15);
            int a = 16 * (l15 + T[15]), b = (5 * Hs + 32) >> 6, c = (5 * Vs + 32) >> 6;
            int base = a + c * (l - 7) + 16;
#pragma unroll
            for (int i = 0; i < 16; i++) out[i] = clip1((base + b * (i - 7)) >> 5);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) out[i] = clip1(out[i] + rs[i]);
        o0 = out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24); o1 = out[4] | (out[5] << 8) | (out[6] << 16) | (out[7] << 24);
        o2 = out[8] | (out[9] << 8) | (out[10] << 16) | (out[11] << 24); o3 = out[12] | (out[13] << 8) | (out[14] << 16) | (out[15] << 24);
    } else if (modes & MBM_T8X8) {
        // ---- Intra8x8 (8.3.2): four 8x8 blocks in order.  Round 5 (intra8_packed.h): a lane reads the block's surroundings from the work tile with thirteen
        //      unconditional dword loads, keeps the filtered reference path (8.3.2.2.1) in seven registers and takes the taps of its four samples (one row
        //      segment of the block) out with v_perm_b32 selectors from a table by (mode, lane): one LDS round trip per block where there were five ----
        uint8_t *tile = lds.ltile(g);
        short *res = lds.lres(g);
        *(uint4 *)(res + l * 16) = res0; *(uint4 *)(res + l * 16 + 8) = res1;
        tile[(1 + l) * kTS + kTO] = (uint8_t)left;
        if (l < 4) ((uint32_t *)(tile + kTO + 1))[l] = ((const uint32_t *)ring_up)[l];
        if (l == 4 || l == 5) ((uint32_t *)(tile + kTO + 17))[l - 4] = ((const uint32_t *)ring_ur)[l - 4];
        if (l == 0) tile[kTO] = (uint8_t)corner;
        const pk::I8Sel *seltab = (const pk::I8Sel *)lds.i8sel();
        const uint32_t m0 = rec[4];                                   // Intra8x8PredMode of block b in nibble b
        const int y8 = l >> 1, x8 = (l & 1) * 4;
#pragma unroll
        for (int b8 = 0; b8 < 4; b8++) {
            const int bx8 = b8 & 1, by8 = b8 >> 1;
            const bool a = bx8 || availA, b = by8 || availB;
            const bool d = (bx8 && by8) ? true : (bx8 ? availB : (by8 ? availA : availD));
            const bool c = b8 == 0 ? availB : (b8 == 1 ? availC : b8 == 2);
            int mode = (m0 >> (4 * b8)) & 15;
            mode = mode > 8 ? 8 : mode;                              // (a damaged record must not index beyond the table)
            const uint4 *sp = (const uint4 *)&seltab[mode * 16 + l];
            const uint4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
            const uint8_t *row0 = tile + (by8 * 8) * kTS + bx8 * 8;   // the dword whose last byte is the block's corner sample; the row above follows
            pk::I8Edge e;
            e.tl = *(const uint32_t *)row0; e.t0 = *(const uint32_t *)(row0 + 4); e.t1 = *(const uint32_t *)(row0 + 8); e.r0 = *(const uint32_t *)(row0 + 12);
            e.r1 = *(const uint32_t *)(row0 + 16);
#pragma unroll
            for (int i = 0; i < 8; i++) e.l[i] = *(const uint32_t *)(row0 + (1 + i) * kTS);      // byte 3 = the left neighbour of the block's row i
            uint32_t F[7];
            pk::i8_filtered_path(e, a, b, c, d, F);
            const int dc = pk::i8_dc(F, a, b);
            pk::I8Sel sel;
            sel.s[0][0] = s0.x; sel.s[0][1] = s0.y; sel.s[0][2] = s0.z; sel.s[0][3] = s0.w; sel.s[1][0] = s1.x; sel.s[1][1] = s1.y; sel.s[1][2] = s1.z;
            sel.s[1][3] = s1.w; sel.s[2][0] = s2.x; sel.s[2][1] = s2.y; sel.s[2][2] = s2.z; sel.s[2][3] = s2.w; sel.m1 = s3.x; sel.m2 = s3.y; sel.m3 = s3.z;
            sel.pad = 0;
            const uint32_t pred = pk::i8_predict4(F, sel, dc);
            const uint2 r = *(const uint2 *)(res + (by8 * 8 + y8) * 16 + bx8 * 8 + x8);       // (issued with the loads above: one wait)
            *(uint32_t *)(tile + (by8 * 8 + 1 + y8) * kTS + kTO + 1 + bx8 * 8 + x8) = pk::add_residual4(pred, r.x, r.y);
        }
        { const uint32_t *rw = (const uint32_t *)(tile + (1 + l) * kTS + kTO + 1); o0 = rw[0]; o1 = rw[1]; o2 = rw[2]; o3 = rw[3]; }
    } else {
        // ---- Intra4x4: work tile in LDS (layout: kTS / kTO above) ----
        // Round 4.  The sixteen blocks are a dependency chain (each predicts from the blocks before it), and rounds 1-3 spent most of a step here:
        // per block three CONDITIONAL byte loads (each its own exec-mask region with its own wait: three LDS round trips in a row), eight more for
        // the DC sum and the address arithmetic of the edge path.  Now a lane fetches its block's whole edge path with seven unconditional aligned
        // dword loads (ONE wait), puts it into four registers and takes its three taps out with two v_perm_b32 whose selectors come from a table
        // (i4_sel_entry); availability is a substitution on the path (unavailable -> 128, no samples above right -> the last one above repeated),
        // decided at compile time for the inner blocks; sums of four samples are one v_sad_u8.
        uint8_t *tile = lds.ltile(g);
        short *res = lds.lres(g);
        *(uint4 *)(res + l * 16) = res0; *(uint4 *)(res + l * 16 + 8) = res1;
        tile[(1 + l) * kTS + kTO] = (uint8_t)left;
        if (l < 4) ((uint32_t *)(tile + kTO + 1))[l] = ((const uint32_t *)ring_up)[l];
        if (l == 4) *(uint32_t *)(tile + kTO + 17) = *(const uint32_t *)ring_ur;
        if (l == 0) tile[kTO] = (uint8_t)corner;
        const uint2 *sel = (const uint2 *)lds.i4sel();
        uint32_t m0 = rec[4], m1 = rec[5];                         // u.i4[8]: two 4-bit modes per byte, raster order
        int px = l & 3, py = l >> 2;
        // everything that does not depend on earlier blocks is fetched up front: selectors and this lane's residuals (two per register)
        uint32_t selA[16], selB[16], rsd2[8];
#pragma unroll
        for (int blk = 0; blk < 16; blk++) {
            const int bx = (blk & 1) + 2 * ((blk >> 2) & 1), by = ((blk >> 1) & 1) + 2 * (blk >> 3), rpos = by * 4 + bx;
            int mode = ((rpos < 8 ? m0 : m1) >> ((rpos & 7) * 4)) & 15;
            mode = mode > 8 ? 8 : mode;                              // (a damaged record must not index beyond the table)
            const uint2 e = sel[mode * 16 + l];
            selA[blk] = e.x; selB[blk] = e.y;
            const uint32_t r = (uint16_t)res[(by * 4 + py) * 16 + bx * 4 + px];
            if (blk & 1) rsd2[blk >> 1] |= r << 16; else rsd2[blk >> 1] = r;
        }
#pragma unroll
        for (int blk = 0; blk < 16; blk++) {
            const int bx = (blk & 1) + 2 * ((blk >> 2) & 1), by = ((blk >> 1) & 1) + 2 * (blk >> 3);
            const bool a = bx > 0 || availA, b = by > 0 || availB;
            const bool cavail = by == 0 ? (bx < 3 ? availB : availC) : !(bx == 3 || blk == 3 || blk == 11 || blk == 7 || blk == 13 || blk == 15);
            // Samples of a neighbour that is not available count as 128, as in k_recon_intra and the oracle.  A conforming stream never selects a
            // mode that reads them; the sweep's generator did (constrained_intra_pred: Horizontal-Down next to an inter corner), and the three
            // decoders have to agree on such a stream too.
            const bool d = (bx > 0 && by > 0) ? true : (bx > 0 ? availB : (by > 0 ? availA : availD));
            const uint8_t *org = tile + (by * 4) * kTS + bx * 4;      // the dword whose last byte is the block's corner sample; the row above follows
            const uint32_t wTL = *(const uint32_t *)org, wT = *(const uint32_t *)(org + 4), wTR = *(const uint32_t *)(org + 8);
            const uint32_t w0 = *(const uint32_t *)(org + kTS), w1 = *(const uint32_t *)(org + 2 * kTS), w2 = *(const uint32_t *)(org + 3 * kTS),
                w3 = *(const uint32_t *)(org + 4 * kTS);              // byte 3 = the left neighbour of the block's rows 0..3
            const uint32_t k128 = 0x80808080u;
            const uint32_t T = b ? wT : k128;
            const uint32_t TR = b ? (cavail ? wTR : __builtin_amdgcn_perm(0u, wT, 0x03030303u)) : k128;
            // P0 = L3 L3 L2 L1 | P1 = L0 TL T0 T1 | P2 = T2 T3 T4 T5 | P3 = T6 T7 T7 -
            uint32_t P0 = __builtin_amdgcn_perm(w1, __builtin_amdgcn_perm(w2, w3, 0x0c070303u), 0x07020100u);
            P0 = a ? P0 : k128;
            const uint32_t lt = __builtin_amdgcn_perm(d ? wTL : k128, a ? w0 : k128, 0x0c0c0703u);          // L0 TL - -
            const uint32_t P1 = lt | (T << 16);
            const uint32_t P2 = __builtin_amdgcn_alignbyte(TR, T, 2), P3 = __builtin_amdgcn_perm(0u, TR, 0x0c030302u);
            const uint32_t X = __builtin_amdgcn_perm(P1, P0, selA[blk]) | __builtin_amdgcn_perm(P3, P2, selB[blk]);
            const int v0 = (int)(X & 255u), v1 = (int)((X >> 8) & 255u), v2 = (int)((X >> 16) & 255u), kd = (int)(selA[blk] >> 24);
            const int st = (int)__builtin_amdgcn_sad_u8(T, 0u, 0u), sl = (int)__builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(lt, P0, 0x04030201u), 0u, 0u);
            const int dc = (a && b) ? (st + sl + 4) >> 3 : (a ? (sl + 2) >> 2 : (b ? (st + 2) >> 2 : 128));
            const int p3 = (v0 + 2 * v1 + v2 + 2) >> 2, p2 = (v1 + v2 + 1) >> 1;
            const int pred = kd == 3 ? dc : (kd == 2 ? p3 : (kd == 1 ? p2 : v1));
            const int r = (int)(short)(rsd2[blk >> 1] >> (16 * (blk & 1)));
            tile[(by * 4 + 1 + py) * kTS + kTO + 1 + bx * 4 + px] = (uint8_t)clip1(pred + r);
        }
        { const uint32_t *rw = (const uint32_t *)(tile + (1 + l) * kTS + kTO + 1); o0 = rw[0]; o1 = rw[1]; o2 = rw[2]; o3 = rw[3]; }
    }
    istore4<WT>(dst, make_uint4(o0, o1, o2, o3));
    rcol[l] = (uint8_t)(o3 >> 24);
    if (l == 15) *(uint4 *)ring_dn = make_uint4(o0, o1, o2, o3);
}

// ------------------------------------------------------------------------------------------
// chroma, one macroblock, 16 lanes: lane = (plane, row).  res: 8 int16 of this lane's row; right2/bottom:
// prefetched samples of an already reconstructed macroblock (UV pair of column 7 in row r; whole row 7).
// ------------------------------------------------------------------------------------------
template <bool WT>
__device__ __forceinline__ void intra_chroma_mb(const ICtx &pp, const ILds &lds, int x, int row, int lrow, int l, int g,
                                uint32_t recdw, uint4 res, uint32_t right2, uint4 bottom) {
    uint8_t *rcol = lds.crcol(lrow);               // [8 rows][2 planes]
    uint8_t *ring_dn = lds.cring(lrow, x & 3);     // 16 B interleaved bottom row
    uint32_t *rec = (uint32_t *)lds.rec(g);
    if (l < 8) rec[l] = recdw;
    uint32_t r0 = rec[0];
    int kind = r0 & 255, modes = (r0 >> 16) & 255, flags = r0 >> 24;
    int plane = l >> 3, r = l & 7;
    if (kind != MB_I4 && kind != MB_I16) {
        rcol[r * 2 + plane] = (uint8_t)(right2 >> (16 + 8 * plane));      // bytes 14, 15 of the row = column 7 (U, V)
        if (l < 4) ((uint32_t *)ring_dn)[l] = l == 0 ? bottom.x : (l == 1 ? bottom.y : (l == 2 ? bottom.z : bottom.w));
        return;
    }
    bool availA = flags & MBF_AVAIL_A, availB = flags & MBF_AVAIL_B;
    int cmode = modes & 3;
    const uint8_t *ring_up = row > 0 ? lds.cring(lrow - 1, x & 3) : ring_dn;
    const uint8_t *ring_ul = row > 0 ? lds.cring(lrow - 1, (x - 1) & 3) : ring_dn;
    int rs[8];
    { uint32_t w[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
      for (int i = 0; i < 8; i++) rs[i] = (int)(short)(w[i >> 1] >> ((i & 1) * 16)); }
    int left = rcol[r * 2 + plane], corner = ring_ul[14 + plane];
    uint4 tv = *(const uint4 *)ring_up;
    int T[8];
    { uint32_t w[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
      for (int i = 0; i < 8; i++) T[i] = (w[i >> 1] >> (((i & 1) * 2 + plane) * 8)) & 255; }
    int out[8];
    if (cmode == 0) {
        int by = r >> 2;
        int sl = sum4(left);                                 // left sum of this lane's 4-row band (same plane: lanes differ in bits 0,1)
        int st0 = T[0] + T[1] + T[2] + T[3], st1 = T[4] + T[5] + T[6] + T[7];
        int dc0, dc1;                                        // blocks (0,by) and (1,by)
        if (by == 0) {
            dc0 = (availA && availB) ? (st0 + sl + 4) >> 3 : (availA ? (sl + 2) >> 2 : (availB ? (st0 + 2) >> 2 : 128));
            dc1 = availB ? (st1 + 2) >> 2 : (availA ? (sl + 2) >> 2 : 128);
        } else {
            dc0 = availA ? (sl + 2) >> 2 : (availB ? (st0 + 2) >> 2 : 128);
            dc1 = (availA && availB) ? (st1 + sl + 4) >> 3 : (availA ? (sl + 2) >> 2 : (availB ? (st1 + 2) >> 2 : 128));
        }
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = i < 4 ? dc0 : dc1;
    } else if (cmode == 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = left;
    } else if (cmode == 2) {
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = T[i];
    } else {
        int Hs = -4 * corner;
#pragma unroll
        for (int i = 0; i < 8; i++) Hs += (i - 3) * T[i];
        int Vs = sum8((r - 3) * left) - 4 * corner;
        int l7 = __shfl(left, (threadIdx.x & 56) | 7);
        int a = 16 * (l7 + T[7]), b = (34 * Hs + 32) >> 6, c = (34 * Vs + 32) >> 6;
        int base = a + c * (r - 3) + 16;
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = clip1((base + b * (i - 3)) >> 5);
    }
    uint8_t *tile = lds.ctile(g);
#pragma unroll
    for (int i = 0; i < 8; i++) { out[i] = clip1(out[i] + rs[i]); tile[r * 16 + 2 * i + plane] = (uint8_t)out[i]; }
    rcol[r * 2 + plane] = (uint8_t)out[7];
    if (l < 8) {
        uint4 v = *(const uint4 *)(tile + l * 16);
        istore4<WT>(pp.plane + (size_t)(row * 8 + l) * pp.pitch + x * 16, v);
        if (l == 7) *(uint4 *)ring_dn = v;
    }
}

// ------------------------------------------------------------------------------------------
// Banded lockstep wavefront, the same construction as k_deblock_band (deblock_lds.hip): a plane is cut into bands of kIBandRows macroblock
// rows, one 4-wave workgroup per band and plane, every band walks the steps s = x + 2 * row of its own rows.  The only coupling is downwards:
// the first row of a band predicts from the bottom sample row of the macroblocks above it.  Those 16 bytes per macroblock travel through the
// picture surface (they are final there anyway) as agent-scope relaxed atomics, behind a per-band step counter in device memory; the band below
// fetches the row of macroblock x + 2 one step before it needs it.  (The first form walked a whole plane with ONE workgroup: up to five
// macroblock rows per 16-lane group at 4K, 7 ms for a 4K I picture.)
constexpr int kIBandRows = 16;
constexpr int kIntraSmemBytes = kTileBase + 32 * 992 + (kIBandRows + 1) * 80 + 64;
// One 256-thread workgroup predicts + reconstructs band `band` of one plane of picture pp.  prog_pic = the picture's band step counters of the
// bottom-row hand-over (kChainIntraRing).  CHAIN (k_chain): the residuals and the samples of non-intra macroblocks come from reconstruction waves of
// the same launch (wait for their bits, cache-bypassing loads), the reconstructed samples are written through, and the band publishes in
// cpic[kChainIntraFin ..] how many of its steps are in memory -- the picture's deblocking bands follow it a few steps behind.
template <bool CHAIN>
__device__ __forceinline__ void intra_band_body(const PicParams &pp, int band, bool is_chroma, int *prog_pic, uint8_t *smem, int *cpic, int *err_word) {
    const int mb_w = pp.mb_w, mb_h = pp.mb_h, pitch = pp.pitch;
    const int row0 = band * kIBandRows;
    if (row0 >= mb_h) return;
    const int rows = min(kIBandRows, mb_h - row0);
    int *prog = prog_pic + (is_chroma ? kDeblockMaxBands : 0);
    const gbyte *resid = (const gbyte *)pp.resid;
    const gbyte *mbs = (const gbyte *)pp.mbs;
    ILds lds{smem, kIBandRows + 1};
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const bool active = g < rows;
    const int row = row0 + (active ? g : 0), lrow = g + 1;
    if (threadIdx.x < 144) { lds.i4tab()[threadIdx.x] = (uint8_t)i4_table_entry(threadIdx.x >> 4, threadIdx.x & 3, (threadIdx.x >> 2) & 3);
        ((uint2 *)lds.i4sel())[threadIdx.x] = i4_sel_entry(threadIdx.x >> 4, threadIdx.x & 3, (threadIdx.x >> 2) & 3); }
    if (!is_chroma && threadIdx.x < 144) ((pk::I8Sel *)lds.i8sel())[threadIdx.x] = pk::i8_sel_entry(threadIdx.x >> 4, threadIdx.x & 15);
    gbyte *plane = (gbyte *)(cur_plane(pp) + (is_chroma ? pp.chroma_offset : 0));
    const ICtx cx{plane, pitch};
    const int rows_per_mb = is_chroma ? 8 : 16, my_row = is_chroma ? (l & 7) : l;
    const bool takes_ring = band > 0 && g == 0;                             // first row of a lower band
    const bool gives_ring = active && g == rows - 1 && row < mb_h - 1;       // last row of a band that has a band below
    const int s_begin = 2 * row0, s_end = mb_w - 1 + 2 * (row0 + rows - 1);
    // bottom sample row of macroblock xm in the row above this band / in this band's last row: 16 bytes, dword `l` by lane l < 4
    const gbyte *above = plane + (size_t)(row0 * rows_per_mb - 1) * pitch + (l & 3) * 4;
    gbyte *below = plane + (size_t)((row + 1) * rows_per_mb - 1) * pitch + (l & 3) * 4;
    int known = 0;
    int *abort_word_ = CHAIN ? cpic - (size_t)pp.chain_idx * kChainStride + (size_t)kChainMaxPics * kChainStride : nullptr;
    auto wait_above = [&](int need) {                                       // (protocol: see k_deblock_band)
        if (band == 0 || threadIdx.x >= 64 || known >= need) return;
        int spins = 0; WaitClock t0;
        while ((known = __hip_atomic_load(&prog[band - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need && ++spins < kSpinLimit && !(CHAIN &&
            wait_expired(spins, t0, abort_word_))) __builtin_amdgcn_s_sleep(8);
        if (CHAIN && l == 0) note_gaps(t0, err_word ? err_word - pp.chain_idx + kChainMaxPics : nullptr);
        // never silent: the engine reports a decode error (and the band does not wait again)
        if (known < need) { if (l == 0) { report_wait_timeout(err_word, CHAIN_ERR_INTRA_TIMEOUT);
            if (CHAIN && abort_word_ && !ld_coh(abort_word_)) { record_first_giveup(abort_word_, CHAIN_ERR_INTRA_TIMEOUT, pp.chain_idx,
                band << 16 | (is_chroma ? 1 : 0),
                need, known, 0);
                record_giveup_evidence(abort_word_, &prog[band - 1], spins, t0); } } known = 0x7fffffff; }
        asm volatile("" ::: "memory");
    };
    auto ring0 = [&](int xm) -> uint32_t * { return (uint32_t *)(is_chroma ? lds.cring(0, xm & 3) : lds.lring(0, xm & 3)); };
    auto fetch_above = [&](int xm) -> uint32_t { return __hip_atomic_load((const JM_GLOBAL uint32_t *)(above + (size_t)xm * 16), __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT); };
    if (band > 0) {
        // macroblocks 0 and 1 of the row above are final once the band above completed step s_begin - 1
        wait_above(s_begin);
        if (takes_ring && l < 8 && (l >> 2) < mb_w) ring0(l >> 2)[l & 3] = fetch_above(l >> 2);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // CHAIN: the residual of an intra macroblock / the samples of a macroblock that is not intra are written by a reconstruction wave of this launch
    int *abort_word = CHAIN ? cpic - (size_t)pp.chain_idx * kChainStride + (size_t)kChainMaxPics * kChainStride : nullptr;
    // PS_RECON beside PS_CHAIN: residuals / samples come from the stage kernel, complete before this launch
    int recon_known = (pp.stages & PS_RECON) ? 0x7fffffff : 0;
    const uint32_t *bits_row = CHAIN ? (const uint32_t *)(cpic + kChainBits) + (size_t)row * kChainRowWords : nullptr;
    const int wave = threadIdx.x >> 6;
    const int wave_first = 2 * (row0 + 4 * wave), wave_last = 2 * (row0 + min(4 * wave + 3, rows - 1)) + mb_w - 1;   // steps in which this wave has work
    uint32_t p_rec = 0, p_right = 0, p_ring = 0; uint4 p_res0 = make_uint4(0, 0, 0, 0), p_res1 = p_res0, p_bot = p_res0;
    // (a macro, not a lambda: capturing the uint4 prefetch registers by reference put them in scratch memory)
#define JM_PREFETCH(s_) do { \
        const int xn_ = (s_) - 2 * row; \
        if (CHAIN && !wait_row_bit(bits_row, recon_known, active && xn_ >= 0 && xn_ < mb_w, xn_, abort_word, pp.chain_idx << 16 | 0x8000 | (row & 0x7fff)) && \
            (threadIdx.x & 63) == 0) { \
            report_wait_timeout(err_word, CHAIN_ERR_BITS_TIMEOUT); st_coh(abort_word, 1); } \
        if (active && xn_ >= 0 && xn_ < mb_w) { \
            const int mb_ = row * mb_w + xn_; \
            p_rec = gload1(mbs + (size_t)mb_ * sizeof(MbRec) + (l & 7) * 4); \
            const gbyte *px_ = plane + (size_t)(row * rows_per_mb) * pitch + xn_ * 16; \
            const gbyte *rs_ = resid + (size_t)mb_ * 768; \
            if (CHAIN) { \
                p_right = gload1_coh(px_ + (size_t)my_row * pitch + 12); \
                p_bot = gload4_coh(px_ + (size_t)(rows_per_mb - 1) * pitch); \
                if (is_chroma) p_res0 = gload4_coh(rs_ + 512 + l * 16); \
                else { p_res0 = gload4_coh(rs_ + l * 32); p_res1 = gload4_coh(rs_ + l * 32 + 16); } \
            } else { \
            p_right = gload1(px_ + (size_t)my_row * pitch + 12); \
            p_bot = gload4(px_ + (size_t)(rows_per_mb - 1) * pitch); \
            if (is_chroma) p_res0 = gload4(rs_ + 512 + l * 16);              /* plane (l >> 3), row (l & 7): 8 int16 */ \
            else { p_res0 = gload4(rs_ + l * 32); p_res1 = gload4(rs_ + l * 32 + 16); } \
            } \
        } \
    } while (0)
    if (s_begin == 2 * row) JM_PREFETCH(s_begin);
    for (int s = s_begin; s <= s_end; s++) {
        const uint32_t c_rec = p_rec, c_right = p_right, c_ring = p_ring; const uint4 c_res0 = p_res0, c_res1 = p_res1, c_bot = p_bot;
        const int x = s - 2 * row;
        // the row above: macroblock x + 1 was fetched during the previous step, macroblock x + 2 is fetched now for the next one (it is final
        // once the band above completed step s, i.e. published s + 1)
        if (takes_ring && l < 4 && x >= 1 && x + 1 < mb_w) ring0(x + 1)[l] = c_ring;
        wait_above(s + 1);
        if (takes_ring && l < 4 && x >= 0 && x + 2 < mb_w) p_ring = fetch_above(x + 2);
        JM_PREFETCH(s + 1);
        if (active && x >= 0 && x < mb_w) {
            if (is_chroma) intra_chroma_mb<CHAIN>(cx, lds, x, row, lrow, l, g, c_rec, c_res0, c_right, c_bot);
            else intra_luma_mb<CHAIN>(cx, lds, x, row, lrow, l, g, c_rec, c_res0, c_res1, c_right, c_bot);
            if (gives_ring) {
                // hand the bottom row down (write-through), then publish the step; the wave that holds the band's last row wrote it itself
                if (l < 4) __hip_atomic_store((JM_GLOBAL uint32_t *)(below + (size_t)x * 16), ((const uint32_t *)(is_chroma ? lds.cring(lrow,
                    x & 3) : lds.lring(lrow, x & 3)))[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (gives_ring) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (l == 0) __hip_atomic_store(&prog[band], s == s_end ? 0x7fffffff : s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (CHAIN) {
            // `ifin`: steps whose samples are in memory, published two steps late (see `fin` in deblock_device.h).  While a wave has work in steps s and
            // s + 1 it has issued the two prefetches (>= 4 loads each) since its stores of step s - 2, so "at most 8 outstanding" proves those stores
            // acknowledged; in the first / last steps of its rows it waits for everything.
            if (s >= wave_first && s + 1 <= wave_last) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (threadIdx.x == 0 && s - 1 > s_begin) __hip_atomic_store(cpic + kChainIntraFin + (is_chroma ? 32 : 0) + band, s - 1, __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
        } else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#undef JM_PREFETCH
    if (CHAIN) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_store(cpic + kChainIntraFin + (is_chroma ? 32 : 0) + band, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace jmamd
