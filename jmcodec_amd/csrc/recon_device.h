// jmcodec_amd/csrc/recon_device.h -- device code of the inter reconstruction (motion compensation + residual), shared by
// k_recon_inter (kernels.hip: one launch per stage, pictures of a batch independent of each other) and k_chain (chain.hip: consecutive
// pictures of one stream in the same launch, coupled by progress counters in device memory).
//
// Part of the replacement for cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:33-41).
#pragma once
#include <hip/hip_runtime.h>
#include "jobs.h"
#include "kernel_common.h"
#include "chain_common.h"
#include "mc_packed.h"

namespace jmamd {

// LDS residual scratch of one wave: 16x16 luma + 2 x 8x8 chroma, int16
struct ResTile { short y[256]; short c[2][64]; int t8[256]; };    // t8: row-transformed 8x8 blocks between the two 1-D passes

// 8.5.13 one-dimensional 8-point inverse transform (in place)
__device__ __forceinline__ void idct8_1d(int *d) {
    int e0 = d[0] + d[4], e1 = -d[3] + d[5] - d[7] - (d[7] >> 1), e2 = d[0] - d[4], e3 = d[1] + d[7] - d[3] - (d[3] >> 1);
    int e4 = (d[2] >> 1) - d[6], e5 = -d[1] + d[7] + d[5] + (d[5] >> 1), e6 = d[2] + (d[6] >> 1), e7 = d[3] + d[5] + d[1] + (d[1] >> 1);
    int f0 = e0 + e6, f1 = e1 + (e7 >> 2), f2 = e2 + e4, f3 = e3 + (e5 >> 2), f4 = e2 - e4, f5 = (e3 >> 2) - e5, f6 = e0 - e6, f7 = e7 - (e1 >> 2);
    d[0] = f0 + f7; d[1] = f2 + f5; d[2] = f4 + f3; d[3] = f6 + f1; d[4] = f6 - f1; d[5] = f4 - f3; d[6] = f2 - f5; d[7] = f0 - f7;
}
// normAdjust8x8(m, i, j) (8.5.9, v_m0..v_m5), six 6-bit fields per m; LevelScale8x8 = weight * this
__device__ __forceinline__ int norm_adjust8(int m, int i, int j) {
    const unsigned long long packed = m == 0 ? 0x6194E0494ull :
        (m == 1 ? 0x69C5634D6ull : (m == 2 ? 0x7E162A5DAull : (m == 3 ? 0x8636AD65Cull : (m == 4 ? 0x9A87B3720ull : 0xAEE8BA824ull))));
    int ti = (i & 1) ? 1 : ((i & 2) ? 2 : 0), tj = (j & 1) ? 1 : ((j & 2) ? 2 : 0);
    int cls = ti == tj ? ti : (ti + tj == 1 ? 3 : (ti + tj == 2 ? 4 : 5));
    return (int)((packed >> (6 * cls)) & 63);
}

// Residual of one macroblock into LDS.  Luma: four lanes per 4x4 block (blkIdx order) -- or lanes 0..31: (8x8 block, row / column)
// when the macroblock uses the 8x8 transform; then lanes 0..31: the eight chroma blocks, four lanes each.
// For MB_I16 the luma DC path (8.5.10) is applied.  Must be called by all 64 lanes of the wave.
__device__ void mb_residual_to_lds(const PicParams &pp, const MbRec &r, ResTile &rt, int lane) {
    const short *coef = pp.coef + r.coef_off;
    int qp = r.qp;
    int n_luma = __popc((unsigned)r.cbp_blk);
    int base_luma = r.kind == MB_I16 ? 16 : 0;
    const bool flat = pp.flat_scaling != 0;
    const int wl = r.kind == MB_INTER ? 3 : 0;                 // scaling list of this macroblock's luma; chroma lists follow it
    if (r.modes & MBM_T8X8) {
        if (lane < 32) {
            int b8 = lane >> 3, i = lane & 7;
            bool coded = (r.cbp_blk >> (4 * b8)) & 1;
            int d[8];
            if (coded) {
                const short *c = coef + 16 * __popc((unsigned)r.cbp_blk & ((1u << (4 * b8)) - 1)) + i * 8;
                int m = qp % 6, s = qp / 6;
#pragma unroll
                for (int k = 0; k < 8; k++) { int wgt = flat ? 16 : pp.wscale8[r.kind == MB_INTER ? 1 : 0][i * 8 + k];
                    int v = c[k] * wgt * norm_adjust8(m, i, k); d[k] = s >= 6 ? v << (s - 6) : (v + (1 << (5 - s))) >> (6 - s); }
                idct8_1d(d);
#pragma unroll
                for (int k = 0; k < 8; k++) rt.t8[b8 * 64 + i * 8 + k] = d[k];
            }
            // second pass: lane = (block, column); same wave, LDS keeps program order
            if (coded) {
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = rt.t8[b8 * 64 + k * 8 + i];
                idct8_1d(d);
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = (d[k] + 32) >> 6;
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = 0;
            }
            int ox = (b8 & 1) * 8 + i, oy = (b8 >> 1) * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) rt.y[(oy + k) * 16 + ox] = (short)d[k];
        }
    } else {
        // Round 5: FOUR lanes per 4x4 block (lane -> block lane >> 2 in coding order, row q of the block), all 64 lanes busy.  Rounds 1-4 gave a block to
        // ONE lane (16 lanes busy, 16 two-byte loads, the whole 2-D transform and 16 two-byte LDS stores in that lane): ~200 wave-instructions for the
        // luma blocks and as many again for the eight chroma blocks on 8 lanes.  A lane now loads its row with one 8-byte load, scales it, runs the row
        // pass of 8.5.12.2 on it, hands the four 32-bit intermediates to the block's other lanes through LDS (rt.t8, unused without the 8x8 transform)
        // and runs the column pass on column q.
        const int blk = lane >> 2, q = lane & 3, rpos = blk_to_raster(blk), bx = rpos & 3, by = rpos >> 2;
        const bool coded = (r.cbp_blk >> blk) & 1;
        int d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        if (coded) {
            const short *c = coef + base_luma + 16 * __popc((unsigned)r.cbp_blk & ((1u << blk) - 1)) + 4 * q;
            const uint2 cw = gld_u2(c);                                               // (coef_off and every block size are multiples of four levels)
            const int c0 = (int)(short)(cw.x & 0xffffu), c1 = (int)cw.x >> 16, c2 = (int)(short)(cw.y & 0xffffu), c3 = (int)cw.y >> 16;
            if (flat) {
                // LevelScale4x4 = 16 * normAdjust4x4: 8.5.12.1 then gives exactly c * (normAdjust << (qP / 6)) for every qP (no rounding term survives)
                const int s = qp / 6, m = qp % 6;
                const int me = ((q & 1) ? norm4(m, 2) : norm4(m, 0)) << s, mo = ((q & 1) ? norm4(m, 1) : norm4(m, 2)) << s;
                d0 = c0 * me; d1 = c1 * mo; d2 = c2 * me; d3 = c3 * mo;
            } else {
                const uint8_t *w4 = &pp.wscale4[wl][4 * q];
                d0 = dequant4w(c0, qp, 4 * q, w4[0]); d1 = dequant4w(c1, qp, 4 * q + 1, w4[1]); d2 = dequant4w(c2, qp, 4 * q + 2, w4[2]);
                d3 = dequant4w(c3, qp, 4 * q + 3, w4[3]);
            }
        }
        bool any = coded;
        if (r.kind == MB_I16) {
            // 8.5.10: f = H c H over the 4x4 DC matrix (H the symmetric Hadamard matrix rows ++++ ++-- +--+ +-+-); element (row by, column bx) belongs to
            // this lane's block.  f[by][bx] = sum over rows q of H[by][q] * (c[q][.] . H[.][bx]): lane q of the block takes row q of c (one 8-byte load) and
            // the four lanes add their terms across the quad (two DPP adds) -- not 16 loads and the whole matrix in every lane.
            const uint2 v = gld_u2(coef + 4 * q);
            const int a = (int)(short)(v.x & 0xffffu), b = (int)v.x >> 16, cc = (int)(short)(v.y & 0xffffu), e = (int)v.y >> 16;
            const int t = bx == 0 ? a + b + cc + e : (bx == 1 ? a + b - cc - e : (bx == 2 ? a - b - cc + e : a - b + cc - e));
            const bool neg = by == 0 ? false : (by == 1 ? q >= 2 : (by == 2 ? (q == 1 || q == 2) : (q & 1) != 0));
            int g = neg ? -t : t;
            g += __builtin_amdgcn_update_dpp(0, g, 0xB1, 0xf, 0xf, false);          // quad_perm [1 0 3 2]
            g += __builtin_amdgcn_update_dpp(0, g, 0x4E, 0xf, 0xf, false);          // quad_perm [2 3 0 1]
            int ls0 = flat ? level_scale4(qp % 6, 0) : pp.wscale4[0][0] * norm4(qp % 6, 0), s = qp / 6;
            const int dc = s >= 6 ? (g * ls0) << (s - 6) : (g * ls0 + (1 << (5 - s))) >> (6 - s);
            if (q == 0) d0 = dc;
            any = true;
        }
        if (any) {
            int *t = &rt.t8[blk * 16];
            {   // rows first (8.5.12.2): this lane's row q
                const int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
                *(int4 *)(t + 4 * q) = make_int4(e0 + e3, e1 + e2, e1 - e2, e0 - e3);
            }
            __builtin_amdgcn_wave_barrier();          // (the four lanes of a block are in one wave, whose LDS operations complete in order)
            const int a = t[q], b = t[4 + q], c = t[8 + q], e = t[12 + q];
            const int g0 = a + c, g1 = a - c, g2 = (b >> 1) - e, g3 = b + (e >> 1);
            short *y = &rt.y[by * 64 + bx * 4 + q];
            y[0] = (short)((g0 + g3 + 32) >> 6); y[16] = (short)((g1 + g2 + 32) >> 6); y[32] = (short)((g1 - g2 + 32) >> 6); y[48] = (short)((g0 - g3 + 32) >> 6);
        } else *(uint2 *)&rt.y[(by * 4 + q) * 16 + bx * 4] = make_uint2(0u, 0u);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 32) {
        // chroma: four lanes per block again -- lane -> (plane, block k4 of the plane, row q)
        const int cb8 = lane >> 2, q = lane & 3, pl = cb8 >> 2, k4 = cb8 & 3, bx = k4 & 1, by = k4 >> 1;
        const int qpc = chroma_qp(qp, pl ? pp.cr_qp_off : pp.cb_qp_off);
        const short *cdc = coef + base_luma + 16 * n_luma;
        const int has_cb = (r.flags & MBF_CB_DC) ? 1 : 0, has_cr = (r.flags & MBF_CR_DC) ? 1 : 0;
        const short *cac = cdc + 4 * (has_cb + has_cr);
        const bool coded = (r.cbp_cac >> cb8) & 1;
        int d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        if (coded) {
            const uint2 cw = gld_u2(cac + 16 * __popc((unsigned)r.cbp_cac & ((1u << cb8) - 1)) + 4 * q);
            const int c0 = (int)(short)(cw.x & 0xffffu), c1 = (int)cw.x >> 16, c2 = (int)(short)(cw.y & 0xffffu), c3 = (int)cw.y >> 16;
            if (flat) {
                const int s = qpc / 6, m = qpc % 6;
                const int me = ((q & 1) ? norm4(m, 2) : norm4(m, 0)) << s, mo = ((q & 1) ? norm4(m, 1) : norm4(m, 2)) << s;
                d0 = c0 * me; d1 = c1 * mo; d2 = c2 * me; d3 = c3 * mo;
            } else {
                const uint8_t *w4 = &pp.wscale4[wl + 1 + pl][4 * q];
                d0 = dequant4w(c0, qpc, 4 * q, w4[0]); d1 = dequant4w(c1, qpc, 4 * q + 1, w4[1]); d2 = dequant4w(c2, qpc, 4 * q + 2, w4[2]);
                d3 = dequant4w(c3, qpc, 4 * q + 3, w4[3]);
            }
        }
        const bool has_dc = pl ? has_cr : has_cb;
        if (q == 0) d0 = 0;                                                          // (level [0] of an AC block is unused: the DC comes from the 2x2 matrix)
        if (has_dc) {
            const uint2 v = gld_u2(cdc + (pl ? 4 * has_cb : 0));
            const int c0 = (int)(short)(v.x & 0xffffu), c1 = (int)v.x >> 16, c2 = (int)(short)(v.y & 0xffffu), c3 = (int)v.y >> 16;
            const int f = k4 == 0 ? c0 + c1 + c2 + c3 : (k4 == 1 ? c0 - c1 + c2 - c3 : (k4 == 2 ? c0 + c1 - c2 - c3 : c0 - c1 - c2 + c3));
            const int dc = ((f * (flat ? level_scale4(qpc % 6, 0) : pp.wscale4[wl + 1 + pl][0] * norm4(qpc % 6, 0))) << (qpc / 6)) >> 5;
            if (q == 0) d0 = dc;
        }
        if (coded || has_dc) {
            int *t = &rt.t8[cb8 * 16];
            {
                const int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
                *(int4 *)(t + 4 * q) = make_int4(e0 + e3, e1 + e2, e1 - e2, e0 - e3);
            }
            __builtin_amdgcn_wave_barrier();
            const int a = t[q], b = t[4 + q], c = t[8 + q], e = t[12 + q];
            const int g0 = a + c, g1 = a - c, g2 = (b >> 1) - e, g3 = b + (e >> 1);
            short *y = &rt.c[pl][by * 32 + bx * 4 + q];
            y[0] = (short)((g0 + g3 + 32) >> 6); y[8] = (short)((g1 + g2 + 32) >> 6); y[16] = (short)((g1 - g2 + 32) >> 6); y[24] = (short)((g0 - g3 + 32) >> 6);
        } else *(uint2 *)&rt.c[pl][(by * 4 + q) * 8 + bx * 4] = make_uint2(0u, 0u);
    }
}

// dynamically indexed fields of a MbRec held in registers: select with shifts instead of indexing an array (which would
// force the record into scratch memory -- 18 MB of extra HBM writes per 1080p picture, measured with WRITE_SIZE)
__device__ __forceinline__ int rec_ref(const MbRec &r, int b8) {
    uint32_t w = (uint32_t)(uint8_t)r.ref[0] | ((uint32_t)(uint8_t)r.ref[1] << 8) | ((uint32_t)(uint8_t)r.ref[2] << 16) | ((uint32_t)(uint8_t)r.ref[3] << 24);
    return (int)(int8_t)(w >> (8 * b8));
}
__device__ __forceinline__ void rec_mv8(const MbRec &r, int b8, int &mx, int &my) {
    uint32_t w0 = (uint16_t)r.u.mv[0][0] | ((uint32_t)(uint16_t)r.u.mv[0][1] << 16), w1 = (uint16_t)r.u.mv[1][0] | ((uint32_t)(uint16_t)r.u.mv[1][1] << 16);
    uint32_t w2 = (uint16_t)r.u.mv[2][0] | ((uint32_t)(uint16_t)r.u.mv[2][1] << 16), w3 = (uint16_t)r.u.mv[3][0] | ((uint32_t)(uint16_t)r.u.mv[3][1] << 16);
    uint32_t w = b8 == 0 ? w0 : (b8 == 1 ? w1 : (b8 == 2 ? w2 : w3));
    mx = (int)(int16_t)(w & 0xffff); my = (int)(int16_t)(w >> 16);
}
__device__ __forceinline__ bool mb_has_residual(const MbRec &r) {
    return r.kind == MB_I16 || r.cbp_blk || r.cbp_cac || (r.flags & (MBF_CB_DC | MBF_CR_DC));
}

// ------------------------------------------------------------------------------------------
// k_recon_inter: one wave per macroblock, 4 macroblocks per workgroup
// ------------------------------------------------------------------------------------------
template <bool COH> __device__ __forceinline__ int ref_luma(const RefBuf &rb, const uint8_t *s, int pitch, int W, int H, int x, int y) {
    return ld_ref8<COH>(rb, &s[clip3(0, H - 1, y) * pitch + clip3(0, W - 1, x)]);     // 32-bit index arithmetic (pointer adds here cost 15 VGPRs)
}
// 8.4.2.2.1 luma sample interpolation for ONE sample, with coordinate clamping (windows that leave the picture, sub-8x8 partitions, missing windows).
// Round 5: one rolled loop over the six rows -- every position is made of G, H, M and the six-tap sums b1 (this row / the row below), h1 (this column / the
// next) and j1, all of which fall out of one pass; the literal form held 36 samples in flight and its 71 registers, added to what is live across the
// call, decided the register count (and the occupancy) of the whole kernel.
template <bool COH> __device__ __noinline__ int luma_sample(const uint8_t *surf_base, const uint8_t *s, int pitch, int W, int H, int xi, int yi, int fx,
    int fy) {
    const RefBuf rb(surf_base);
    int G = 0, Hs = 0, M = 0, b0 = 0, b1 = 0, h0 = 0, h1 = 0, j1 = 0;
#pragma unroll 1
    for (int j = 0; j < 6; j++) {
        const int y = yi + j - 2;
        const int p0 = ref_luma<COH>(rb, s, pitch, W, H, xi - 2, y), p1 = ref_luma<COH>(rb, s, pitch, W, H, xi - 1, y), p2 = ref_luma<COH>(rb, s, pitch, W, H, xi, y);
        const int p3 = ref_luma<COH>(rb, s, pitch, W, H, xi + 1, y), p4 = ref_luma<COH>(rb, s, pitch, W, H, xi + 2, y), p5 = ref_luma<COH>(rb, s, pitch, W, H, xi + 3, y);
        const int hb = tap6(p0, p1, p2, p3, p4, p5);
        const int tap = (j == 0 || j == 5) ? 1 : ((j == 1 || j == 4) ? -5 : 20);
        j1 += tap * hb; h0 += tap * p2; h1 += tap * p3;
        if (j == 2) { G = p2; Hs = p3; b0 = hb; }
        if (j == 3) { M = p2; b1 = hb; }
    }
    if (!fx && !fy) return G;
    const int b = clip1(((fy == 3 ? b1 : b0) + 16) >> 5);          // b, or s = b of the row below
    const int h = clip1(((fx == 3 ? h1 : h0) + 16) >> 5);          // h, or m = h of the next column
    if (!fy) return fx == 2 ? b : (((fx == 1 ? G : Hs) + b + 1) >> 1);
    if (!fx) return fy == 2 ? h : (((fy == 1 ? G : M) + h + 1) >> 1);
    if (fx == 2 || fy == 2) {
        const int jj = clip1((j1 + 512) >> 10);
        if (fx == 2 && fy == 2) return jj;
        return ((fx == 2 ? b : h) + jj + 1) >> 1;
    }
    return (b + h + 1) >> 1;
}

// LDS of one 4-wave workgroup of the inter reconstruction
constexpr int kUniChroma = 136;          // dword offset of the chroma window behind the 21 x 6 luma window in wins[wave] (uni path)
// Chain launches, "quad" path (round 5): when the four macroblocks of a workgroup all take the one-window path from the same reference picture with vectors
// that lie close together, the workgroup fetches ONE window for the four of them -- kQRows rows of kQStride dwords of luma, kQCRows of chroma -- instead of
// four overlapping ones (coherent loads are served from memory in 64-byte requests whoever read the line before: four private 24-byte rows cost 5.2 requests, one
// shared 72-byte row 2.1; profiles/r05_chain_quad_fetch.txt).
constexpr int kQStride = pk::kQuadStride, kQRows = pk::kQuadRows, kQCRows = pk::kQuadChromaRows;      // (mc_packed.h: quad_geometry, host-tested)
struct alignas(16) ReconLds {
    ResTile tiles[4];
    uint32_t outt[4][96];                        // per wave: reconstructed MB, 16 luma rows + 8 interleaved chroma rows of 16 B
    union {
        uint32_t wins[4][4][13 * 5 + 3];         // per wave, per 8x8 block: 13 rows x 5 dwords of reference window
        struct { uint32_t qwin[kQRows * kQStride]; uint32_t qcwin[kQCRows * kQStride]; };      // the workgroup's shared window (quad path)
    };
    uint32_t fate[4];                            // chain launches: what became of each wave's macroblock (the store tail of recon_inter_wave)
    int vote[4][8];                              // chain launches: each wave's one-window parameters (reference slot or -1, xi, yi, 2 * cxi, cyi)
};

// One wave reconstructs macroblock `mb` (mbx, mby) of picture pp.  CHAIN = false: the stage kernel (every reference picture was complete
// before the launch).  CHAIN = true: the chain kernel -- reference pictures may be decoded by EARLIER workgroups of the same launch, so the
// wave first waits until the deblocking of everything its reference windows touch is final (ChainView::wait_final), reads reference
// samples with loads that bypass the non-coherent cache levels, writes its result through to memory and publishes the macroblock in the
// picture's reconstruction bitmap, on which the deblocking workgroups of this picture wait.
// COH: the reference loads are the cache-bypassing kind (needed exactly when the picture has references inside the launch, pp.n_deps > 0).
// HAS_BI: macroblocks with two-list / weighted motion records take their luma windows through LDS like P blocks when they can (below).  The stage
// kernel has an instantiation without it for batches that hold no such picture: the extra code costs 9 VGPRs there (96 -> 105, 5 -> 4 waves per SIMD).
// FIELD = false: no picture of the launch is a field picture -- a reference entry is a plain surface index and the picture its whole surface (round 4:
// what the frame-only kernels of round 2 compiled to; k_recon_inter<false> needs its 96 registers for five waves per SIMD, kernels.hip)
template <bool CHAIN, bool COH, bool HAS_BI = true, bool FIELD = true>
__device__ __forceinline__ void recon_inter_wave(const PicParams &pp, int mb, bool valid, ReconLds &sm, const ChainView &cv) {
    // (frame pictures: the surface's address by arithmetic -- q.surf[slot] is a load that depends on the record, one more round trip in front of the window loads)
    auto ref_plane = [&](const PicParams &q, int slot) -> const uint8_t * { return FIELD ? jmamd::ref_plane(q, slot) : q.surf_base + (size_t)slot * q.surf_stride; };
    auto chroma_mvy_offset = [&](const PicParams &q, int slot) -> int { return FIELD ? jmamd::chroma_mvy_offset(q, slot) : 0; };
    auto cur_plane = [&](const PicParams &q) -> uint8_t * { return FIELD ? jmamd::cur_plane(q) : q.surf_base + (size_t)q.cur * q.surf_stride; };
    const RefBuf refbuf(pp.surf_base);                           // the coherent reference loads of chain launches (chain_common.h)
    ResTile *tiles = sm.tiles;
    uint32_t (*outt)[96] = sm.outt;
    uint32_t (*wins)[4][13 * 5 + 3] = sm.wins;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Round 5: `mb` and `valid` are wave-uniform AND the compiler is told so (the callers pass them through v_readfirstlane): the record comes in with
    // scalar loads and stays in scalar registers, every branch on it is a scalar branch.  Before, the record sat in vector registers, each of the ~40
    // conditions on it cost an exec-mask save / restore pair (SQ_INSTS_SALU: 289 per macroblock) and mb % mb_w / mb / mb_w ran as two 25-instruction
    // reciprocal sequences per lane.
    if (!CHAIN && !valid) return;                               // (the stage kernel has no workgroup barrier behind this point)
    const MbWords rw = load_mbrec_uniform(pp.mbs + (valid ? mb : 0));
    MbRec r = __builtin_bit_cast(MbRec, rw);
    if (!valid) { r.kind = MB_I4; r.cbp_blk = 0; r.cbp_cac = 0; r.flags = 0; }
    bool inter = valid && r.kind == MB_INTER;
    bool intra_res = valid && pp.want_intra_resid && (r.kind == MB_I4 || r.kind == MB_I16);
    bool has_res = (inter || intra_res) && mb_has_residual(r);
    const int mby = valid ? (pp.mb_w_magic ? (int)__umulhi((uint32_t)mb, pp.mb_w_magic) : mb) : 0, mbx = valid ? mb - mby * pp.mb_w : 0;   // mb / mb_w (jobs.h)
    const int W = pp.mb_w * 16, H = pp.mb_h * 16, pitch = pp.pitch;
    // Everything an ordinary inter macroblock reads from its reference pictures depends only on the record, not on the residual:
    // issue those loads first so that their latency overlaps the coefficient loads and the inverse transform (the kernel is
    // latency bound: SQ_WAIT_ANY was 65 % of SQ_WAVE_CYCLES with the loads issued where they were consumed).
    // HAS_BI = false: NO picture of the launch has two-list / weighted motion records (Engine: any_bipred; the parser sets MBM_BIPRED only in slices that
    // have them), so that instantiation is compiled without their code -- round 5: it was the register high-water mark of every instantiation (97 against
    // 70 without it: five waves per SIMD against seven)
    const bool bipred = HAS_BI && inter && (r.modes & MBM_BIPRED);
    // (an instantiation without the two-list code that meets such a record all the same -- a future producer that breaks the invariant above -- says so
    //  through the picture's error word instead of decoding the record as a plain one; the picture is then reported as damaged, never silently wrong)
    // (reported at the END of the wave's work: a store to memory the compiler knows nothing about, placed here, turned every later scalar load of the picture's
    //  parameters and of the record into a vector load -- + 5 % vector instructions per macroblock, profiles/r06_sq_counters_c1.json)
    const bool bad_record = !HAS_BI && inter && (r.modes & MBM_BIPRED);
    const bool plain = inter && !bipred;
    bool fast = false;
    uint32_t wv[5] = {0, 0, 0, 0, 0};                       // this lane's dwords of the 13x13 reference window (fast path)
    uint32_t c_wa = 0, c_wb = 0;                            // chroma: the four neighbours of (cx, cy) as U V U V of its row and of the row below
    int c_slot = -1, c_fx = 0, c_fy = 0;
    if (CHAIN && COH) {
        // lane -> (4x4 block rb, list): the samples that block's prediction can touch (13x13-style window: -2 .. +3 around the block, which
        // also covers the chroma samples of the same vector) must be final in the reference picture before anything is read from it
        const int rb = lane >> 2, list = (lane >> 1) & 1, bx = rb & 3, by = rb >> 2, b8 = (by >> 1) * 2 + (bx >> 1);
        int slot = -1, mvx = 0, mvy = 0;
        if (inter) {
            if (bipred) {
                const short *rec = pp.mv_ext + (size_t)r.u.mv_ext * 2;
                mvx = rec[list * 32 + rb * 2]; mvy = rec[list * 32 + rb * 2 + 1];
                slot = list ? ((const int8_t *)(rec + 64))[b8] : rec_ref(r, b8);
            } else if (list == 0) {
                if (r.flags & MBF_MV_EXT) { const short *m = pp.mv_ext + ((size_t)r.u.mv_ext + rb) * 2; mvx = m[0]; mvy = m[1]; }
                else rec_mv8(r, b8, mvx, mvy);
                slot = rec_ref(r, b8);
            }
        }
        const int dep = slot >= 0 ? (int)pp.dep_pic[slot & 31] : -1;
        const int x0 = mbx * 16 + bx * 4 + (mvx >> 2), y0 = mby * 16 + by * 4 + (mvy >> 2);
        const uint32_t tw0 = cv.census_on ? (uint32_t)wall_clock64() : 0u;
        const bool ok = cv.wait_final(dep, clip3(0, W - 1, x0 + 6), clip3(0, H - 1, y0 - 2), clip3(0, H - 1, y0 + 6), pp.mb_w, pp.mb_h, kRowLag);
        if (cv.census_on) cv.census(ChainView::CENSUS_WAIT_TICKS, (int)((uint32_t)wall_clock64() - tw0));
        if (!ok && lane == 0) report_wait_timeout(cv.err + pp.chain_idx, CHAIN_ERR_FIN_TIMEOUT);
    }
    // Round 4: ONE window per macroblock when its four 8x8 blocks share a vector and a reference (P_Skip, P_L0_16x16: 97 % of the inter macroblocks of
    // config C1): 21 rows of 6 aligned dwords for luma and 9 rows of 5 for the interleaved chroma, three loads per lane instead of nine -- the four 13x13
    // windows overlap by half, and the chroma path fetched four bytes per sample pair.  Measured (profiles/r04_ab9_one_window.json): k_recon_inter 534 ->
    // 504 us per launch; the bytes the chain kernels fetch from memory moved by 3 % only (what they are: profiles/r04_pmc_chain_l2_requests.txt).
    // `uni` is wave-uniform (the record is).
    bool uni = false;
    uint32_t cw = 0;                                        // this lane's dword of the chroma window (uni path)
    int c_sh = 0;                                           // byte offset of chroma sample 0 in its window row
    int u_fx = 0, u_fy = 0, u_sh = 0;                       // uni path: the luma vector's fraction and the byte offset of window column 0 in its dword
    int u_xi = 0, u_yi = 0, u_cx2 = 0, u_cyi = 0;           // uni path: first sample of the luma window, first byte / row of the chroma window
    int qv_rows = 0, qv_dws = 0, qc_rows = 0, qc_dws = 0;   // quad path: extent of the shared windows
    if (plain && !(r.flags & MBF_MV_EXT) && rec_ref(r, 0) >= 0) {
        // (the record as dwords: four vectors and the four reference bytes compared with a handful of scalar operations, not field by field)
        const uint32_t m0 = rw.w[4], refs = rw.w[3];
        const bool same = ((rw.w[5] ^ m0) | (rw.w[6] ^ m0) | (rw.w[7] ^ m0) | (refs ^ ((refs & 255u) * 0x01010101u))) == 0;
        if (same) {
            int mvx = (int16_t)(m0 & 0xffff), mvy = (int16_t)(m0 >> 16);
            const int slot = rec_ref(r, 0);
            const int xi = mbx * 16 + (mvx >> 2) - 2, yi = mby * 16 + (mvy >> 2) - 2;
            const int cmvy = mvy + chroma_mvy_offset(pp, slot);
            const int cxi = mbx * 8 + (mvx >> 3), cyi = mby * 8 + (cmvy >> 3);
            // the luma window inside the picture; in a frame picture that puts the chroma window (half the vector, 9 x 9 samples) inside as well:
            // cxi == (xi + 2) >> 1 >= 1 and cxi + 9 <= (W - 19) / 2 + 9 < W / 2, rows alike, and its five dwords end at most at byte W <= pitch
            uni = W >= 21 && H >= 21 && (unsigned)xi <= (unsigned)(W - 21) && (unsigned)yi <= (unsigned)(H - 21);
            if (FIELD) uni = uni && cyi >= 0 && cyi + 9 <= (H >> 1);                            // (the chroma vector of a cross-parity reference is shifted by 2)
            if (uni) {
                u_xi = xi; u_yi = yi; u_cx2 = 2 * cxi; u_cyi = cyi;
                c_sh = (2 * cxi) & 3;
                c_slot = slot; c_fx = mvx & 7; c_fy = cmvy & 7;
                u_fx = mvx & 3; u_fy = mvy & 3; u_sh = xi & 3;
            }
        }
    }
    // Chain launches: can the workgroup's four macroblocks share one window?  Every wave says what its window would be; all four decide alike.
    bool quad = false;
    int q_row = 0, q_crow = 0;                              // quad path: this macroblock's first row in the shared luma / chroma window
    uint32_t qv[4] = {0, 0, 0, 0}, qc[2] = {0, 0};          // quad path: this lane's dwords of the shared windows
    if (CHAIN && COH) {
        if (lane == 0) { int *v = sm.vote[wave]; v[0] = uni ? c_slot : -1; v[1] = u_xi; v[2] = u_yi; v[3] = u_cx2; v[4] = u_cyi; }
        __syncthreads();
        const pk::QuadGeom qg = pk::quad_geometry(sm.vote);
        quad = qg.ok;
        cv.census(quad ? ChainView::CENSUS_QUAD : ChainView::CENSUS_PRIVATE);      // (diagnostic launches only: JM_AMD_DEC_CENSUS)
        const int x0 = qg.x0, y0 = qg.y0, cx0 = qg.cx0, cy0 = qg.cy0;
        if (quad) {
            // rows wave * 2 + (lane >> 5) + 8 k of the shared window, dword lane & 31 of the row: unconditional loads (clamped), stored where they belong below
            const uint8_t *ref = ref_plane(pp, c_slot);
            const int nrow = qg.nrow, ndw = qg.ndw, ncrow = qg.ncrow, ncdw = qg.ncdw;
            const int sub = wave * 2 + (lane >> 5), dwi = lane & 31;
#pragma unroll
            for (int k = 0; k < 4; k++) qv[k] = ld_ref32<COH>(refbuf, ref, (uint32_t)((y0 + min(sub + 8 * k, nrow - 1)) * pitch + x0 + min(dwi, ndw - 1) * 4));
            const uint8_t *rc = ref + pp.chroma_offset;
#pragma unroll
            for (int k = 0; k < 2; k++)
                qc[k] = ld_ref32<COH>(refbuf, rc, (uint32_t)((cy0 + min(sub + 8 * k, ncrow - 1)) * pitch + cx0 + min(dwi, ncdw - 1) * 4));
            q_row = u_yi - y0; q_crow = u_cyi - cy0;
            u_sh = u_xi - x0; c_sh = u_cx2 - cx0;           // byte offsets of this macroblock's first sample in a row of the shared windows
            // (what the stores below need: the window's extent, in registers the loop above already holds)
            qv_rows = nrow; qv_dws = ndw; qc_rows = ncrow; qc_dws = ncdw;
        }
    }
    if (uni && !quad) {
        const uint8_t *ref = ref_plane(pp, c_slot);
        const int xa = u_xi & ~3;
#pragma unroll
        // (unconditional loads -- lanes past the window's end fetch its last dword again: a load under `if (i < 126)` came out as its own block
        // with its own s_waitcnt vmcnt(0), i.e. the two loads of a lane were two memory round trips one after the other)
        for (int t = 0; t < 2; t++) { const int i = min(lane + 64 * t, 125);
            wv[t] = ld_ref32<COH>(refbuf, ref, (uint32_t)((u_yi + i / 6) * pitch + xa + (i % 6) * 4)); }
        const uint8_t *rc = ref + pp.chroma_offset;
        const int ca = u_cx2 & ~3;
        { const int i = min(lane, 44); cw = ld_ref32<COH>(refbuf, rc, (uint32_t)((u_cyi + i / 5) * pitch + ca + (i % 5) * 4)); }
    }
    if (plain && !uni) {
        {
            int g = lane >> 4;                              // 8x8 block of this lane in the fast-path mapping
            int mvx, mvy; rec_mv8(r, g, mvx, mvy);
            int xi = mbx * 16 + (g & 1) * 8 + (mvx >> 2) - 2, yi = mby * 16 + (g >> 1) * 8 + (mvy >> 2) - 2;
            bool ok = !(r.flags & MBF_MV_EXT) && rec_ref(r, g) >= 0 && xi >= 0 && yi >= 0 && xi + 13 <= W && yi + 13 <= H;
            fast = __all(ok);                               // wave-uniform: the whole macroblock takes one path
            if (fast) {
                const uint8_t *ref = ref_plane(pp, rec_ref(r, g));
                int l = lane & 15, xa = xi & ~3;
#pragma unroll
                for (int t = 0; t < 5; t++) {
                    const int i = min(l + 16 * t, 64), row = i / 5, dw = i % 5;
                    wv[t] = ld_ref32<COH>(refbuf, ref, (uint32_t)((yi + row) * pitch + xa + dw * 4));
                }
            }
        }
        {
            int cx = lane & 7, cy = lane >> 3;
            int rb = (cy >> 1) * 4 + (cx >> 1), b8 = (cy >> 2) * 2 + (cx >> 2);
            int mvx, mvy;
            if (r.flags & MBF_MV_EXT) { const short *m = pp.mv_ext + ((size_t)r.u.mv_ext + rb) * 2; mvx = m[0]; mvy = m[1]; }
            else rec_mv8(r, b8, mvx, mvy);
            c_slot = rec_ref(r, b8);
            if (c_slot >= 0) mvy += chroma_mvy_offset(pp, c_slot);
            c_fx = mvx & 7; c_fy = mvy & 7;
            if (c_slot >= 0) {
                const uint8_t *rc = ref_plane(pp, c_slot) + pp.chroma_offset;
                int CW = W >> 1, CHh = H >> 1;
                int xi = mbx * 8 + cx + (mvx >> 3), yi = mby * 8 + cy + (mvy >> 3);
                int xa = clip3(0, CW - 1, xi), xb = clip3(0, CW - 1, xi + 1), ya = clip3(0, CHh - 1, yi), yb = clip3(0, CHh - 1, yi + 1);
                const uint8_t *r0 = rc + (size_t)ya * pitch, *r1 = rc + (size_t)yb * pitch;
                if (COH && __all(xb == xa + 1 && 2 * xa + 8 <= pitch)) {
                    // chain launches: the four bytes U V U V at 2 * xa of each row come out of two aligned dwords per row -- cache-bypassing BYTE loads
                    // are one memory request each (FETCH_SIZE showed 26.9 MB per 1080p picture for k_chain against 4.9 MB for the stage kernels)
                    const int o = (2 * xa) & ~3, sh = (2 * xa) & 3;
                    const uint32_t a0 = ld_ref32<true>(refbuf, r0 + o), a1 = ld_ref32<true>(refbuf, r0 + o + 4), b0 = ld_ref32<true>(refbuf, r1 + o),
                        b1 = ld_ref32<true>(refbuf, r1 + o + 4);
                    c_wa = __builtin_amdgcn_alignbyte(a1, a0, sh); c_wb = __builtin_amdgcn_alignbyte(b1, b0, sh);
                } else {
                    // a U V pair per 16-bit load (the pair starts at an even byte); xa / xb are clamped separately at the picture's edge
                    c_wa = ld_ref16<COH>(refbuf, r0 + 2 * xa) | ld_ref16<COH>(refbuf, r0 + 2 * xb) << 16;
                    c_wb = ld_ref16<COH>(refbuf, r1 + 2 * xa) | ld_ref16<COH>(refbuf, r1 + 2 * xb) << 16;
                }
            }
        }
    }
    if (has_res) mb_residual_to_lds(pp, r, tiles[wave], lane);
    // the residual tile is private to this wave and LDS operations of one wave complete in order: no workgroup barrier needed
    __builtin_amdgcn_wave_barrier();
    uint8_t *dst = cur_plane(pp);
    uint8_t *dst_c = dst + pp.chroma_offset;
    uint32_t *ot = outt[wave];
    // what the tail below does with this wave's macroblock: nothing (no macroblock), publish it in the picture's reconstruction bitmap (chain launches;
    // what it had to write is in memory), or store its samples out of `ot` first
    enum : uint32_t { FATE_NONE = 0, FATE_PUBLISH = 1, FATE_STORE = 2 };
    uint32_t fate = FATE_NONE;
    do {
    if (!valid) break;
    if (intra_res) {
        // residual of an intra macroblock for k_intra_lds: 384 int16 (Y 16x16, Cb 8x8, Cr 8x8), zeros when nothing is coded
        if (lane < 48) {
            uint4 v = has_res ? *(const uint4 *)((const short *)&tiles[wave] + lane * 8) : make_uint4(0, 0, 0, 0);
            if (CHAIN) st_wt16(pp.resid + (size_t)mb * 384 + lane * 8, v);        // read by the intra band of this launch, possibly on another XCD
            else *(uint4 *)(pp.resid + (size_t)mb * 384 + lane * 8) = v;
        }
        if (CHAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (this wave's own stores: in memory before the bit is)
        fate = FATE_PUBLISH;
        break;
    }
    if (r.kind == MB_PCM) {
        const uint8_t *pcm = (const uint8_t *)(pp.coef + r.coef_off);
        int row = lane >> 2, xq = lane & 3;
        ot[row * 4 + xq] = pcm[row * 16 + xq * 4] | (pcm[row * 16 + xq * 4 + 1] << 8) | (pcm[row * 16 + xq * 4 + 2] << 16) | (pcm[row * 16 + xq * 4 + 3] << 24);
        int cx = lane & 7, cy = lane >> 3;
        ((uint16_t *)(ot + 64))[cy * 8 + cx] = (uint16_t)(pcm[256 + cy * 8 + cx] | (pcm[320 + cy * 8 + cx] << 8));
    }
    if (!inter && r.kind != MB_PCM) { fate = FATE_PUBLISH; break; }
    // Luma prediction from a window in LDS: pk::mc_luma4 (mc_packed.h).  The windows hold the samples ^ 0x80 (signed bytes for v_dot4_i32_i8); a lane
    // filters four neighbouring samples of one row; the fractional position is uniform inside an 8x8 block (and inside the macroblock on the one-window
    // path), so the six-tap paths do not diverge within a block's 16 lanes.
    // stage_window: this lane's five dwords of a block's 13x13 window (row * 5 + dword == l16 + 16 * t) into the block's LDS window
    auto stage_window = [&](uint32_t *win, const uint32_t *w5, int l16) {
#pragma unroll
        for (int t = 0; t < 5; t++) { const int i = l16 + 16 * t; if (i < 65) win[i] = w5[t] ^ pk::kSign; }
    };
    if (bipred) {
        // B slices / weighted prediction: one or two references per 8x8, one vector per 4x4, weights of 8.4.2.3.  Literal sampling.
        const short *rec = pp.mv_ext + (size_t)r.u.mv_ext * 2;
        const int8_t *tail = (const int8_t *)(rec + 64);
        const SliceWp *wp = pp.wp ? &pp.wp[r.slice] : nullptr;
        const int mode = wp ? wp->mode : 0;
        auto combine = [&](int a, int b, bool use0, bool use1, int i0, int i1, int cmp) -> int {
            if (use0 && use1) {
                if (mode == 0) return (a + b + 1) >> 1;
                int w0, w1, o = 0, lg = 5;
                if (mode == 1) { w0 = wp->w[0][i0 & 15][cmp]; w1 = wp->w[1][i1 & 15][cmp]; o = (wp->o[0][i0 & 15][cmp] + wp->o[1][i1 & 15][cmp] + 1) >> 1;
                    lg = cmp ? wp->logwd_c : wp->logwd_y; }
                else { w1 = (int)wp->imp_w1[i0 & 15][i1 & 15] - 64; w0 = 64 - w1; }
                return clip1(((a * w0 + b * w1 + (1 << lg)) >> (lg + 1)) + o);
            }
            if (!use0 && !use1) return 128;
            int v = use0 ? a : b;
            if (mode != 1) return v;
            int l = use0 ? 0 : 1, i = (use0 ? i0 : i1) & 15, lg = cmp ? wp->logwd_c : wp->logwd_y, w = wp->w[l][i][cmp], o = wp->o[l][i][cmp];
            return clip1((lg >= 1 ? ((v * w + (1 << (lg - 1))) >> lg) : v * w) + o);
        };
        // luma.  Fast path: every 8x8 block has ONE vector per list in use (B_16x16 / 16x8 / 8x16, B_8x8 with 8x8 sub-blocks, direct
        // prediction with direct_8x8_inference) and its 13x13 windows lie inside the picture -- then each list's window goes through LDS exactly like a
        // P block's (5 dword loads per lane instead of up to 36 byte loads per SAMPLE) and the two predictions are combined per sample.
        bool bfast = false;
        if (HAS_BI) {
            const int g = lane >> 4, rb0 = (g >> 1) * 8 + (g & 1) * 2;
            bool okb = true;
#pragma unroll
            for (int L = 0; L < 2; L++) {
                if ((L ? (int)tail[g] : rec_ref(r, g)) < 0) continue;
                const short *m = rec + L * 32 + rb0 * 2;
                const int ax = m[0], ay = m[1];
                okb = okb && m[2] == ax && m[3] == ay && m[8] == ax && m[9] == ay && m[10] == ax && m[11] == ay;
                const int xi = mbx * 16 + (g & 1) * 8 + (ax >> 2) - 2, yi = mby * 16 + (g >> 1) * 8 + (ay >> 2) - 2;
                okb = okb && xi >= 0 && yi >= 0 && xi + 13 <= W && yi + 13 <= H;
            }
            bfast = __all(okb);
        }
        if (HAS_BI && bfast) {
            const int g = lane >> 4, l16 = lane & 15, rb0 = (g >> 1) * 8 + (g & 1) * 2;
            const int s0 = rec_ref(r, g), s1 = tail[g], i0 = tail[4 + g], i1 = tail[8 + g];
            uint32_t pa = 0, pb = 0;                               // the two predictions of this lane's four samples, one byte each
            // (round 5: BOTH lists' windows are fetched before either is filtered -- one memory round trip for the macroblock instead of two in a row)
            struct Win { uint32_t w5[5]; int sh, fx, fy; };
            auto fetch = [&](int slot, const short *m, Win &wn) {
                const uint32_t mvw = *(const uint32_t *)m;
                const int mvx = (int16_t)(mvw & 0xffffu), mvy = (int)mvw >> 16;
                const int xi = mbx * 16 + (g & 1) * 8 + (mvx >> 2) - 2, yi = mby * 16 + (g >> 1) * 8 + (mvy >> 2) - 2, xa = xi & ~3;
                const uint8_t *ref = ref_plane(pp, slot);
#pragma unroll
                for (int t = 0; t < 5; t++) { const int i = min(l16 + 16 * t, 64), row = i / 5, dw = i % 5;
                    wn.w5[t] = ld_ref32<COH>(refbuf, ref, (uint32_t)((yi + row) * pitch + xa + dw * 4)); }
                wn.sh = xi & 3; wn.fx = mvx & 3; wn.fy = mvy & 3;
            };
            auto filter = [&](const Win &wn) -> uint32_t {
                __builtin_amdgcn_wave_barrier();                   // (the block's window in LDS is reused for the second list: its readers are done)
                stage_window(&wins[wave][g][0], wn.w5, l16);
                __builtin_amdgcn_wave_barrier();
                const uint32_t pv = pk::mc_luma4(&wins[wave][g][0], 5, l16 >> 1, 4 * (l16 & 1) + wn.sh, wn.fx, wn.fy);
                __builtin_amdgcn_wave_barrier();
                return pv;
            };
            Win wa, wb;
            if (s0 >= 0) fetch(s0, rec + rb0 * 2, wa);
            if (s1 >= 0) fetch(s1, rec + 32 + rb0 * 2, wb);
            if (s0 >= 0) pa = filter(wa);
            if (s1 >= 0) pb = filter(wb);
            const int rr = l16 >> 1, hh = l16 & 1, px = (g & 1) * 8 + hh * 4, py = (g >> 1) * 8 + rr;
            uint32_t pred;
            if (mode == 0) pred = s0 >= 0 ? (s1 >= 0 ? pk::lerp(pa, pb, pk::kOnes) : pa) : (s1 >= 0 ? pb : pk::kSign);   // default: the rounded average, or the one list
            else {
                int v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = combine((int)((pa >> (8 * k)) & 255), (int)((pb >> (8 * k)) & 255), s0 >= 0, s1 >= 0, i0, i1, 0);
                pred = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
            }
            if (has_res) { const uint2 rs = *(const uint2 *)&tiles[wave].y[py * 16 + px]; pred = pk::add_residual4(pred, rs.x, rs.y); }
            ot[py * 4 + (px >> 2)] = pred;
        } else
        {   // luma: lane -> (4x4 block, row)
            int rb = lane >> 2, row = lane & 3, bx = rb & 3, by = rb >> 2, b8 = (by >> 1) * 2 + (bx >> 1);
            int s0 = rec_ref(r, b8), s1 = tail[b8], i0 = tail[4 + b8], i1 = tail[8 + b8];
            int x0 = mbx * 16 + bx * 4, y = mby * 16 + by * 4 + row;
            int m0x = rec[rb * 2], m0y = rec[rb * 2 + 1], m1x = rec[32 + rb * 2], m1y = rec[32 + rb * 2 + 1];
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int a = 0, b = 0;
                if (s0 >= 0) a = luma_sample<COH>(pp.surf_base, ref_plane(pp, s0), pitch, W, H, x0 + k + (m0x >> 2), y + (m0y >> 2), m0x & 3, m0y & 3);
                if (s1 >= 0) b = luma_sample<COH>(pp.surf_base, ref_plane(pp, s1), pitch, W, H, x0 + k + (m1x >> 2), y + (m1y >> 2), m1x & 3, m1y & 3);
                v[k] = combine(a, b, s0 >= 0, s1 >= 0, i0, i1, 0);
            }
            if (has_res) { const short *rs = &tiles[wave].y[(by * 4 + row) * 16 + bx * 4];
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = clip1(v[k] + rs[k]); }
            ot[(by * 4 + row) * 4 + bx] = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
        }
        {   // chroma: lane -> (cx, cy), both planes
            int cx = lane & 7, cy = lane >> 3, rb = (cy >> 1) * 4 + (cx >> 1), b8 = (cy >> 2) * 2 + (cx >> 2);
            int s[2] = { rec_ref(r, b8), tail[b8] }, i0 = tail[4 + b8], i1 = tail[8 + b8];
            // (round 5: a U V pair per 16-bit load and one v_dot4 per plane and list, as in the one-list paths -- the byte-wise form issued sixteen loads per lane)
            int CW = W >> 1, CHh = H >> 1;
            uint32_t puv[2] = {0, 0};                                   // per list: U | V << 8
#pragma unroll
            for (int l = 0; l < 2; l++) {
                if (s[l] < 0) continue;
                const uint32_t mvw = *(const uint32_t *)(rec + l * 32 + rb * 2);
                int mvx = (int16_t)(mvw & 0xffffu), mvy = (int)mvw >> 16;
                mvy += chroma_mvy_offset(pp, s[l]);
                const uint8_t *rc = ref_plane(pp, s[l]) + pp.chroma_offset;
                int xi = mbx * 8 + cx + (mvx >> 3), yi = mby * 8 + cy + (mvy >> 3), fx = mvx & 7, fy = mvy & 7;
                int xa = clip3(0, CW - 1, xi), xb = clip3(0, CW - 1, xi + 1), ya = clip3(0, CHh - 1, yi), yb = clip3(0, CHh - 1, yi + 1);
                const uint8_t *r0 = rc + (size_t)ya * pitch, *r1 = rc + (size_t)yb * pitch;
                const uint32_t wa = ld_ref16<COH>(refbuf, r0 + 2 * xa) | ld_ref16<COH>(refbuf, r0 + 2 * xb) << 16;
                const uint32_t wb = ld_ref16<COH>(refbuf, r1 + 2 * xa) | ld_ref16<COH>(refbuf, r1 + 2 * xb) << 16;
                puv[l] = pk::mc_chroma_uv(wa, wb, pk::chroma_weights(fx, fy));
            }
            int u, v;
            if (mode == 0) { const uint32_t uv = s[0] >= 0 ? (s[1] >= 0 ? pk::lerp(puv[0], puv[1], pk::kOnes) : puv[0]) : (s[1] >= 0 ? puv[1] : 0x8080u);
                u = (int)(uv & 255u); v = (int)((uv >> 8) & 255u); }
            else { u = combine((int)(puv[0] & 255u), (int)(puv[1] & 255u), s[0] >= 0, s[1] >= 0, i0, i1, 1);
                v = combine((int)((puv[0] >> 8) & 255u), (int)((puv[1] >> 8) & 255u), s[0] >= 0, s[1] >= 0, i0, i1, 2); }
            if (has_res) { u = clip1(u + tiles[wave].c[0][cy * 8 + cx]); v = clip1(v + tiles[wave].c[1][cy * 8 + cx]); }
            ((uint16_t *)(ot + 64))[cy * 8 + cx] = (uint16_t)(u | (v << 8));
        }
    } else if (inter) {
    // ---- luma ----
    // Fast path (one MV per 8x8 block, i.e. 16x16 / 16x8 / 8x16 / 8x8 partitions): 16 lanes per 8x8 block stage its
    // 13x13 reference window in LDS with aligned dword loads, then every lane filters 4 pixels of one row out of LDS.
    // The fractional position is uniform inside a block, so the 6-tap paths do not diverge within the 16 lanes.
    // Slow path (sub-8x8 partitions, windows touching the picture border, missing reference): literal per-sample taps.
    uint32_t pred;                                              // this lane's four luma samples (px .. px + 3, py) of the macroblock
    int px, py;
    if (CHAIN && COH && quad) {
        // the workgroup's shared window (every wave of the workgroup is here: `quad` is the same for all four)
        const int sub = wave * 2 + (lane >> 5), dwi = lane & 31;
#pragma unroll
        for (int k = 0; k < 4; k++) if (sub + 8 * k < qv_rows && dwi < qv_dws) sm.qwin[(sub + 8 * k) * kQStride + dwi] = qv[k] ^ pk::kSign;
#pragma unroll
        for (int k = 0; k < 2; k++) if (sub + 8 * k < qc_rows && dwi < qc_dws) sm.qcwin[(sub + 8 * k) * kQStride + dwi] = qc[k];
        __syncthreads();
        py = lane >> 2; px = (lane & 3) * 4;
        pred = pk::mc_luma4(sm.qwin, kQStride, py + q_row, px + u_sh, u_fx, u_fy);
    } else if (uni) {
        // the macroblock's window: 21 rows x 6 dwords (+ the chroma window behind it), written by all 64 lanes, read back by the same wave;
        // lane -> row lane >> 2, samples 4 * (lane & 3) ..: the output tile and the residual are then walked linearly
        uint32_t *w16 = &wins[wave][0][0];
        if (lane < 62) { w16[lane] = wv[0] ^ pk::kSign; w16[lane + 64] = wv[1] ^ pk::kSign; } else w16[lane] = wv[0] ^ pk::kSign;
        if (lane < 45) w16[kUniChroma + lane] = cw;
        __builtin_amdgcn_wave_barrier();
        py = lane >> 2; px = (lane & 3) * 4;
        pred = pk::mc_luma4(w16, 6, py, px + u_sh, u_fx, u_fy);
    } else if (fast) {
        const int g = lane >> 4, l = lane & 15;
        int mvx, mvy; rec_mv8(r, g, mvx, mvy);
        const int xi = mbx * 16 + (g & 1) * 8 + (mvx >> 2) - 2;
        stage_window(&wins[wave][g][0], wv, l);
        __builtin_amdgcn_wave_barrier();
        const int rr = l >> 1, hh = l & 1;
        px = (g & 1) * 8 + hh * 4; py = (g >> 1) * 8 + rr;         // position inside the macroblock
        pred = pk::mc_luma4(&wins[wave][g][0], 5, rr, 4 * hh + (xi & 3), mvx & 3, mvy & 3);
    } else {
        int rb = lane >> 2, row = lane & 3;
        int bx = rb & 3, by = rb >> 2;
        int b8 = (by >> 1) * 2 + (bx >> 1);
        int mvx, mvy;
        if (r.flags & MBF_MV_EXT) { const short *m = pp.mv_ext + ((size_t)r.u.mv_ext + rb) * 2; mvx = m[0]; mvy = m[1]; }
        else rec_mv8(r, b8, mvx, mvy);
        int slot = rec_ref(r, b8);
        int x0 = mbx * 16 + bx * 4, y = mby * 16 + by * 4 + row;
        int v[4];
        if (slot < 0) { v[0] = v[1] = v[2] = v[3] = 128; }
        else {
            const uint8_t *ref = ref_plane(pp, slot);
            int xi = x0 + (mvx >> 2), yi = y + (mvy >> 2), fx = mvx & 3, fy = mvy & 3;
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = luma_sample<COH>(pp.surf_base, ref, pitch, W, H, xi + k, yi, fx, fy);
        }
        px = bx * 4; py = by * 4 + row;
        pred = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    }
    if (has_res) { const uint2 rs = *(const uint2 *)&tiles[wave].y[py * 16 + px]; pred = pk::add_residual4(pred, rs.x, rs.y); }
    ot[py * 4 + (px >> 2)] = pred;
    // ---- chroma: lane -> chroma position (cx, cy), both planes; the four neighbours were loaded up front ----
    {
        const int cx = lane & 7, cy = lane >> 3;
        if (uni) {
            // U V U V at byte c_sh + 2 * cx of window rows cy and cy + 1 (5 dwords per row; the shared window: kQStride): two dwords per row, aligned by byte
            const bool q = CHAIN && COH && quad;
            const uint32_t *cwn = q ? sm.qcwin + q_crow * kQStride : &wins[wave][0][0] + kUniChroma;
            const int cs = q ? kQStride : 5;
            pk::chroma_pairs(cwn, cs, cy, c_sh + 2 * cx, c_wa, c_wb);
        }
        // 8.4.2.2.2: one v_dot4 per plane with the weights (8 - x)(8 - y), x (8 - y), (8 - x) y, x y
        uint32_t uv = c_slot < 0 ? 0x8080u : pk::mc_chroma_uv(c_wa, c_wb, pk::chroma_weights(c_fx, c_fy));
        if (has_res) uv = pk::add_residual_uv(uv, tiles[wave].c[0][cy * 8 + cx], tiles[wave].c[1][cy * 8 + cx]);
        ((uint16_t *)(ot + 64))[cy * 8 + cx] = (uint16_t)uv;
    }
    }   // inter
    fate = FATE_STORE;
    } while (0);
    // ---- store ----
    if (CHAIN) {
        // The four waves of a workgroup hold four macroblocks side by side (x a multiple of 4: 64 bytes of every row, 64-byte aligned).  Stored wave by
        // wave, every write-through row was a 16-byte piece of a line of its own (WRITE_SIZE: 22 MB per 1080p picture for 4.7 MB of samples); wave 0
        // stores all four out of the shared LDS tile instead, four neighbouring lanes a whole 64-byte segment, and publishes the four macroblocks with
        // ONE atomic once its stores are in memory (wave 0's macroblock always exists: the chain kernels drop a workgroup whose first one does not).
        if (lane == 0) sm.fate[wave] = fate;
        __syncthreads();
        if (wave == 0) {
            const int piece = lane & 3, row = lane >> 2;
            const uint32_t f = sm.fate[piece];
            if (f == FATE_STORE) {
                st_wt16(dst + (size_t)(mby * 16 + row) * pitch + (mbx + piece) * 16, *(const uint4 *)(outt[piece] + row * 4));
                if (lane < 32) st_wt16(dst_c + (size_t)(mby * 8 + row) * pitch + (mbx + piece) * 16, *(const uint4 *)(outt[piece] + 64 + row * 4));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t bits = (uint32_t)(__builtin_amdgcn_ballot_w64(lane < 4 && f != FATE_NONE) & 15u) << (mbx & 31);
            if (lane == 0 && bits) __hip_atomic_fetch_or((uint32_t *)(cv.pic(pp.chain_idx) + kChainBits) + mby * kChainRowWords + (mbx >> 5), bits,
                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (fate == FATE_STORE) {
        // whole 16-byte rows (lanes 0..15 luma, 16..23 interleaved chroma), so that HBM sees full segments
        if (lane < 16) gst_u4(dst + (size_t)(mby * 16 + lane) * pitch + mbx * 16, *(const uint4 *)(ot + lane * 4));
        else if (lane < 24) gst_u4(dst_c + (size_t)(mby * 8 + lane - 16) * pitch + mbx * 16, *(const uint4 *)(ot + 64 + (lane - 16) * 4));
    }
    if (bad_record && cv.err && lane == 0) report_wait_timeout(cv.err + blockIdx.y, CHAIN_ERR_BAD_RECORD);
}

}  // namespace jmamd
