// jmcodec_amd/csrc/mc_packed.h -- the sample arithmetic of the inter reconstruction on PACKED bytes / packed 16-bit pairs (round 5).
//
// What it replaces: rounds 1-4 unpacked every byte of a reference window into an int (three v_alignbyte + nine shift / mask per window row) and ran the
// six-tap filter of 8.4.2.2.1 as scalar integer code, the centre positions with six horizontal filters PER OUTPUT SAMPLE; SQ_INSTS_VALU showed 757
// wave-instructions per macroblock for k_recon_inter, 3.6 x what the deblocking spends.  Here a lane keeps its samples four to a register:
//   * the window in LDS holds the samples XOR 0x80, i.e. as SIGNED bytes p - 128, so that a six-tap sum is two v_dot4_i32_i8 (taps 1 -5 20 20 | -5 1 0 0
//     on two byte windows made by v_alignbyte); the taps sum to 32, so 32 * 128 = 4096 is added back through the accumulator (gfx950 has no mixed-sign dot);
//   * vertical sums: v_perm_b32 interleaves two rows (r0[0] r1[0] r0[1] r1[1]), three v_dot4 per sample with the taps in the matching byte pair;
//   * rounding shift + Clip1 of two samples at once: v_pk_ashrrev_i16, v_sat_pk_u8_i16; the quarter-sample averages of four samples: ONE v_lerp_u8;
//   * chroma (8.4.2.2.2): v_perm gathers the four neighbours of one plane, one v_dot4_u32_u8 with the weights (8-x)(8-y) x(8-y) (8-x)y xy;
//   * prediction + residual: bytes -> packed 16, v_pk_add_i16 with clamp, v_sat_pk_u8_i16.
// Every function is __host__ __device__: on the host the seven instructions are restated in plain C++, so that tests/test_mc_packed.py checks the
// arithmetic of every fractional position against a literal restatement of the clause WITHOUT a GPU (tests/native/mc_packed_check.cpp); on the GPU the
// parity tests against the oracle cover the same code with the real instructions.
//
// Part of the replacement for cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:33-41).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define JM_HD __host__ __device__ __forceinline__
#else
#define JM_HD static inline
#endif

namespace jmamd {
namespace pk {

#if defined(__HIP_DEVICE_COMPILE__)
typedef short pk_s2 __attribute__((ext_vector_type(2)));
JM_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }      // ({hi, lo} >> 8 * (sh & 3))
JM_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }             // byte k = byte sel[k] of {hi, lo}
JM_HD int sdot4(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }            // signed bytes
JM_HD uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }            // unsigned bytes
JM_HD uint32_t lerp(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_lerp(a, b, c); }                     // per byte (a + b + (c & 1)) >> 1
JM_HD uint32_t sat_pk_u8(uint32_t a) { uint32_t d; asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(a)); return d; }      // two int16 -> two clamped bytes
JM_HD uint32_t pk_add_sat(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(pk_s2, a), __builtin_bit_cast(pk_s2, b)));
}
JM_HD uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pk_s2, a) + __builtin_bit_cast(pk_s2, b)); }
JM_HD uint32_t pk_ashr(uint32_t a, int n) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pk_s2, a) >> (pk_s2)(short)n); }
#else
// ---- the same seven instructions in plain C++ (GCN3 / CDNA3 instruction set manuals) ----
JM_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3))); }
JM_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t pool = ((uint64_t)hi << 32) | lo;
    uint32_t d = 0;
    for (int k = 0; k < 4; k++) {
        const uint32_t s = (sel >> (8 * k)) & 255;
        uint32_t b;
        if (s <= 7) b = (uint32_t)(pool >> (8 * s)) & 255;
        else if (s <= 11) b = ((pool >> (16 * (s - 8) + 15)) & 1) ? 255 : 0;      // sign of the 16-bit word s - 8
        else if (s == 12) b = 0;
        else b = 255;
        d |= b << (8 * k);
    }
    return d;
}
JM_HD int sdot4(uint32_t a, uint32_t b, int c) {
    for (int k = 0; k < 4; k++) c += (int)(int8_t)(a >> (8 * k)) * (int)(int8_t)(b >> (8 * k));
    return c;
}
JM_HD uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
    for (int k = 0; k < 4; k++) c += ((a >> (8 * k)) & 255) * ((b >> (8 * k)) & 255);
    return c;
}
JM_HD uint32_t lerp(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d = 0;
    for (int k = 0; k < 4; k++) d |= ((((a >> (8 * k)) & 255) + ((b >> (8 * k)) & 255) + ((c >> (8 * k)) & 1)) >> 1) << (8 * k);
    return d;
}
JM_HD uint32_t sat_pk_u8(uint32_t a) {
    const int lo = (int16_t)(a & 0xffff), hi = (int16_t)(a >> 16);
    return (uint32_t)(lo < 0 ? 0 : (lo > 255 ? 255 : lo)) | (uint32_t)(hi < 0 ? 0 : (hi > 255 ? 255 : hi)) << 8;
}
JM_HD uint32_t pk_add_sat(uint32_t a, uint32_t b) {
    int lo = (int16_t)(a & 0xffff) + (int16_t)(b & 0xffff), hi = (int16_t)(a >> 16) + (int16_t)(b >> 16);
    lo = lo < -32768 ? -32768 : (lo > 32767 ? 32767 : lo); hi = hi < -32768 ? -32768 : (hi > 32767 ? 32767 : hi);
    return ((uint32_t)lo & 0xffff) | ((uint32_t)hi << 16);
}
JM_HD uint32_t pk_add(uint32_t a, uint32_t b) { return ((a + b) & 0xffff) | (((a >> 16) + (b >> 16)) << 16); }
JM_HD uint32_t pk_ashr(uint32_t a, int n) {
    return ((uint32_t)((int16_t)(a & 0xffff) >> n) & 0xffff) | ((uint32_t)((int16_t)(a >> 16) >> n) << 16);
}
#endif

constexpr uint32_t kSign = 0x80808080u;          // samples are kept as p ^ 0x80 = (int8)(p - 128) in the luma windows
constexpr uint32_t kTapA = 0x1414FB01u;          // taps  1 -5 20 20 on bytes 0..3 (8.4.2.2.1: 1 -5 20 20 -5 1)
constexpr uint32_t kTapB = 0x000001FBu;          // taps -5  1  0  0
constexpr int kUnbias = 4096;                    // 32 * 128: what the -128 of six signed samples takes from a six-tap sum
constexpr uint32_t kOnes = 0x01010101u;

// bytes cb .. cb + 11 of a window row (cb: any byte offset) as three dwords
struct Row12 { uint32_t a0, a1, a2; };
JM_HD Row12 row12(const uint32_t *row, int cb) {
    const uint32_t *p = row + (cb >> 2);
    const uint32_t s = (uint32_t)cb & 3u, d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3];
    Row12 r; r.a0 = alignbyte(d1, d0, s); r.a1 = alignbyte(d2, d1, s); r.a2 = alignbyte(d3, d2, s);
    return r;
}
// bytes cb .. cb + 3
JM_HD uint32_t row4(const uint32_t *row, int cb) { const uint32_t *p = row + (cb >> 2); return alignbyte(p[1], p[0], (uint32_t)cb & 3u); }

// h[k] = tap6(b[k] .. b[k + 5]) + (bias - 4096) for the nine signed bytes b[0..8] at the front of r, k = 0..3 (pass bias = 4096 + rounding)
JM_HD void hsum4(const Row12 &r, int bias, int *h) {
    const uint32_t w1 = alignbyte(r.a1, r.a0, 1), w2 = alignbyte(r.a1, r.a0, 2), w3 = alignbyte(r.a1, r.a0, 3);
    const uint32_t w5 = alignbyte(r.a2, r.a1, 1), w6 = alignbyte(r.a2, r.a1, 2), w7 = alignbyte(r.a2, r.a1, 3);
    h[0] = sdot4(r.a0, kTapA, sdot4(r.a1, kTapB, bias));
    h[1] = sdot4(w1, kTapA, sdot4(w5, kTapB, bias));
    h[2] = sdot4(w2, kTapA, sdot4(w6, kTapB, bias));
    h[3] = sdot4(w3, kTapA, sdot4(w7, kTapB, bias));
}
// v[k] = tap6(c[0][k] .. c[5][k]) + (bias - 4096): six rows, four neighbouring signed samples each
JM_HD void vsum4(const uint32_t *c, int bias, int *v) {
    const uint32_t t01l = perm(c[1], c[0], 0x05010400u), t01h = perm(c[1], c[0], 0x07030602u);      // r0[0] r1[0] r0[1] r1[1] | r0[2] r1[2] r0[3] r1[3]
    const uint32_t t23l = perm(c[3], c[2], 0x05010400u), t23h = perm(c[3], c[2], 0x07030602u);
    const uint32_t t45l = perm(c[5], c[4], 0x05010400u), t45h = perm(c[5], c[4], 0x07030602u);
    v[0] = sdot4(t01l, 0x0000FB01u, sdot4(t23l, 0x00001414u, sdot4(t45l, 0x000001FBu, bias)));
    v[1] = sdot4(t01l, 0xFB010000u, sdot4(t23l, 0x14140000u, sdot4(t45l, 0x01FB0000u, bias)));
    v[2] = sdot4(t01h, 0x0000FB01u, sdot4(t23h, 0x00001414u, sdot4(t45h, 0x000001FBu, bias)));
    v[3] = sdot4(t01h, 0xFB010000u, sdot4(t23h, 0x14140000u, sdot4(t45h, 0x01FB0000u, bias)));
}
// Clip1(x[k] >> shift) of four values, one byte each.  clip_pack16: every x[k] fits 16 bits BEFORE the shift (single six-tap sums: -2,550 .. 10,710 + 16)
JM_HD uint32_t clip_pack16(const int *x, int shift) {
    const uint32_t lo = pk_ashr(perm((uint32_t)x[1], (uint32_t)x[0], 0x05040100u), shift), hi = pk_ashr(perm((uint32_t)x[3], (uint32_t)x[2], 0x05040100u), shift);
    return perm(sat_pk_u8(hi), sat_pk_u8(lo), 0x05040100u);
}
JM_HD uint32_t clip_pack32(const int *x, int shift) {       // (the centre sums need 20 bits; after >> 10 they fit 16)
    const uint32_t lo = perm((uint32_t)(x[1] >> shift), (uint32_t)(x[0] >> shift), 0x05040100u), hi = perm((uint32_t)(x[3] >> shift), (uint32_t)(x[2] >> shift), 0x05040100u);
    return perm(sat_pk_u8(hi), sat_pk_u8(lo), 0x05040100u);
}

// 8.4.2.2.1 for the four samples (x .. x + 3, y) of one lane, quarter-sample position (fx, fy).
// win: reference window, samples ^ 0x80, `stride` dwords per row; its rows wr .. wr + 5 are the sample rows y - 2 .. y + 3 and the bytes cb .. cb + 8
// of a row the sample columns x - 2 .. x + 6.  Returns the four predicted samples, one byte each (x in byte 0).
JM_HD uint32_t mc_luma4(const uint32_t *win, int stride, int wr, int cb, int fx, int fy) {
    if (fy == 0) {
        const uint32_t *row = win + (wr + 2) * stride;
        if (fx == 0) return row4(row, cb + 2) ^ kSign;
        const Row12 r = row12(row, cb);
        int h[4]; hsum4(r, kUnbias + 16, h);
        const uint32_t b = clip_pack16(h, 5);
        if (fx == 2) return b;
        return lerp(alignbyte(r.a1, r.a0, fx == 1 ? 2u : 3u) ^ kSign, b, kOnes);                 // a: (G + b + 1) >> 1, c: (H + b + 1) >> 1
    }
    if (fx == 0) {
        uint32_t c[6];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < 6; j++) c[j] = row4(win + (wr + j) * stride, cb + 2);
        int v[4]; vsum4(c, kUnbias + 16, v);
        const uint32_t h = clip_pack16(v, 5);
        if (fy == 2) return h;
        return lerp((fy == 1 ? c[2] : c[3]) ^ kSign, h, kOnes);                                  // d: (G + h + 1) >> 1, n: (M + h + 1) >> 1
    }
    if (fx == 2 || fy == 2) {
        // j (and f, q, i, k around it): the six-tap filter over the unrounded horizontal sums of six rows
        // (row by row into the vertical sum: four accumulators instead of a 6 x 4 array of live intermediates -- k_recon_inter must stay within 96 registers)
        int jv[4] = {512, 512, 512, 512}, hq[4] = {0, 0, 0, 0}; uint32_t cc[6];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < 6; j++) {
            const Row12 r = row12(win + (wr + j) * stride, cb);
            int h[4]; hsum4(r, kUnbias, h);
            const int tap = (j == 0 || j == 5) ? 1 : ((j == 1 || j == 4) ? -5 : 20);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int k = 0; k < 4; k++) jv[k] += tap * h[k];
            if (j == 2 || j == 3) {                                                             // b1 of this row (f) or of the row below (q)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int k = 0; k < 4; k++) hq[k] = (fy == 1) == (j == 2) ? h[k] : hq[k];
            }
            cc[j] = alignbyte(r.a1, r.a0, fx == 1 ? 2u : 3u);                                       // (used by i / k only)
        }
        const uint32_t jj = clip_pack32(jv, 10);
        if (fx == 2 && fy == 2) return jj;
        uint32_t q;
        if (fx == 2) {                                                                           // f: b of this row, q: s = b of the row below
            int t[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int k = 0; k < 4; k++) t[k] = hq[k] + 16;
            q = clip_pack16(t, 5);
        } else {                                                                                 // i: h of this column, k: m = h of the next column
            int v[4]; vsum4(cc, kUnbias + 16, v);
            q = clip_pack16(v, 5);
        }
        return lerp(q, jj, kOnes);
    }
    // e, g, p, r: the horizontal half sample of this row or the next (b / s) and the vertical one of this column or the next (h / m)
    const Row12 r = row12(win + (wr + (fy == 1 ? 2 : 3)) * stride, cb);
    int h[4]; hsum4(r, kUnbias + 16, h);
    const uint32_t bq = clip_pack16(h, 5);
    uint32_t c[6];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 6; j++) c[j] = row4(win + (wr + j) * stride, cb + (fx == 1 ? 2 : 3));
    int v[4]; vsum4(c, kUnbias + 16, v);
    return lerp(bq, clip_pack16(v, 5), kOnes);
}

// 8.4.2.2.2: weights of the four neighbours A B C D of a chroma sample at eighth-sample position (fx, fy), one byte each
JM_HD uint32_t chroma_weights(int fx, int fy) {
    return (uint32_t)((8 - fx) * (8 - fy)) | (uint32_t)(fx * (8 - fy)) << 8 | (uint32_t)((8 - fx) * fy) << 16 | (uint32_t)(fx * fy) << 24;
}
// wa / wb: the bytes U V U V at the sample's integer position in its row and in the row below (interleaved chroma plane) -> U | V << 8
JM_HD uint32_t mc_chroma_uv(uint32_t wa, uint32_t wb, uint32_t w) {
    const uint32_t u = udot4(perm(wb, wa, 0x06040200u), w, 32u) >> 6, v = udot4(perm(wb, wa, 0x07050301u), w, 32u) >> 6;
    return u | v << 8;
}
// Clip1(prediction + residual) for four samples: pred one byte each, r01 / r23 the residuals as packed int16 pairs
JM_HD uint32_t add_residual4(uint32_t pred, uint32_t r01, uint32_t r23) {
    const uint32_t lo = pk_add_sat(perm(0u, pred, 0x0c010c00u), r01), hi = pk_add_sat(perm(0u, pred, 0x0c030c02u), r23);
    return perm(sat_pk_u8(hi), sat_pk_u8(lo), 0x05040100u);
}
// the same for one U / V pair: uv = U | V << 8, residuals ru, rv (int16 range)
JM_HD uint32_t add_residual_uv(uint32_t uv, int ru, int rv) {
    return sat_pk_u8(pk_add_sat(perm(0u, uv, 0x0c010c00u), ((uint32_t)ru & 0xffffu) | ((uint32_t)rv << 16))) & 0xffffu;
}

// ---- one reference window for the four macroblocks of a chain workgroup (recon_device.h, "quad" path) ----
// Each macroblock says what its own one-window fetch would be: vote[w] = { reference slot or -1, xi, yi, 2 * cxi, cyi } -- the first luma sample of its 21 x 21
// window, the first byte / row of its 9-row chroma window (interleaved U V).  The shared window starts at (x0, y0) / (cx0, cy0), x0 and cx0 dword-aligned, and
// holds nrow x ndw dwords of luma, ncrow x ncdw of chroma, in rows of kQuadStride dwords.  ok: all four take the one-window path from the same picture and
// everything a macroblock reads of a row (pk::row12: four dwords from the dword of byte 12 + offset at most) lies inside a row of kQuadStride dwords.
constexpr int kQuadStride = 27, kQuadRows = 32, kQuadChromaRows = 16;
struct QuadGeom { bool ok; int x0, y0, nrow, ndw, cx0, cy0, ncrow, ncdw; };
JM_HD QuadGeom quad_geometry(const int (*vote)[8]) {
    int x0 = 1 << 30, x1 = 0, y0 = 1 << 30, y1 = 0, cx0 = 1 << 30, cx1 = 0, cy0 = 1 << 30, cy1 = 0; bool all = true;
    for (int w = 0; w < 4; w++) {
        const int sl = vote[w][0], a = vote[w][1], b = vote[w][2], c = vote[w][3], d = vote[w][4];
        all = all && sl >= 0 && sl == vote[0][0];
        x0 = a < x0 ? a : x0; x1 = a > x1 ? a : x1; y0 = b < y0 ? b : y0; y1 = b > y1 ? b : y1;
        cx0 = c < cx0 ? c : cx0; cx1 = c > cx1 ? c : cx1; cy0 = d < cy0 ? d : cy0; cy1 = d > cy1 ? d : cy1;
    }
    x0 &= ~3; cx0 &= ~3;
    QuadGeom g;
    g.ok = all && x1 - x0 + 28 <= kQuadStride * 4 && y1 - y0 + 21 <= kQuadRows && cx1 - cx0 + 24 <= kQuadStride * 4 && cy1 - cy0 + 9 <= kQuadChromaRows;
    g.x0 = x0; g.y0 = y0; g.nrow = y1 - y0 + 21; g.ndw = (x1 - x0 + 21 + 3) >> 2;
    g.cx0 = cx0; g.cy0 = cy0; g.ncrow = cy1 - cy0 + 9; g.ncdw = (cx1 - cx0 + 18 + 3) >> 2;
    return g;
}
// the four interleaved chroma bytes U V U V at byte offset `off` of window rows r and r + 1 (`stride` dwords per row): what mc_chroma_uv takes
JM_HD void chroma_pairs(const uint32_t *win, int stride, int r, int off, uint32_t &wa, uint32_t &wb) {
    const int o = off >> 2; const uint32_t sh = (uint32_t)off & 3u;
    wa = alignbyte(win[r * stride + o + 1], win[r * stride + o], sh); wb = alignbyte(win[(r + 1) * stride + o + 1], win[(r + 1) * stride + o], sh);
}

}  // namespace pk
}  // namespace jmamd
