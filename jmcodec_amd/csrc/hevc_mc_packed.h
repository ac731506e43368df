// jmcodec_amd/csrc/hevc_mc_packed.h -- the fractional sample interpolation of HEVC (ITU-T H.265 8.5.3.3.3) on PACKED bytes / packed 16-bit pairs (round 6).
//
// What it replaces: rounds 1-5 ran the separable 8-tap / 4-tap filters of k_hevc_mc as plain integer code -- per sample eight byte reads from LDS and eight
// multiply-adds, twice -- 29.7 M VALU wave-instructions per 4K picture (profiles/r05_sq_counters_c3.json), two and a half times what the H.264 kernel spends
// since its round-5 rewrite (mc_packed.h).  The same construction here:
//   * the reference window in LDS holds the samples XOR 0x80, i.e. as SIGNED bytes p - 128 (gfx950 has no mixed-sign dot product), in rows of 32 bytes;
//   * horizontal pass: a lane takes two window rows and four neighbouring columns; an 8-tap sum is two v_dot4_i32_i8 (taps 0..3 | 4..7) on byte windows
//     made by v_alignbyte: 6 alignbytes + 8 dots for four sums.  The sums stay BIASED by -128 * (sum of the taps) and fit 16 bits (8-bit video:
//     -12,272 .. 12,208), and shift1 of the clause is 0 for 8-bit samples, so nothing is rounded between the passes;
//   * the lane packs its two rows into 16-bit pairs (row 2k | row 2k + 1 << 16) and stores them COLUMN-major: a column of the intermediate array is then a
//     string of dwords, and the vertical 8-tap sum of any row is four (even row) or five (odd row) v_dot2_i32_i16 on consecutive dwords -- the taps packed
//     in pairs, shifted by one position for odd rows;
//   * interleaved chroma (Cb Cr Cb Cr): the horizontal 4-tap sum of one component is one v_dot4 on the bytes v_perm picks out of two dwords; vertically the
//     interleaved row is treated like a luma row (the component of a column does not change down the column);
//   * the bias comes back at the end: predSample = (sum + bias) >> shift, one constant pair per (xFrac != 0, yFrac != 0) case.
// Every function is __host__ __device__; on the host the instructions are restated in plain C++ (mc_packed.h), so that tests/test_hevc_mc_packed.py checks every
// fractional position against a literal restatement of the clause without a GPU; on the GPU the HEVC parity tests run the same code with the real instructions.
//
// Part of the replacement for cuvidDecodePicture with codec_type 1 (/root/reference/nv_dec/nv_dec.cpp:33-41; CUVIDHEVCPICPARAMS, nv_sdk/inc/dynlink_cuviddec.h:428-530).
#pragma once
#include "mc_packed.h"

namespace jmamd {
namespace hpk {

using pk::alignbyte; using pk::perm; using pk::sdot4;

#if defined(__HIP_DEVICE_COMPILE__)
typedef short hpk_s2 __attribute__((ext_vector_type(2)));
JM_HD int sdot2(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(hpk_s2, a), __builtin_bit_cast(hpk_s2, b), c, false); }   // v_dot2_i32_i16
#else
JM_HD int sdot2(uint32_t a, uint32_t b, int c) { return c + (int)(int16_t)(a & 0xffff) * (int)(int16_t)(b & 0xffff) + (int)(int16_t)(a >> 16) * (int)(int16_t)(b >> 16); }
#endif

// Table 8-11 (luma, quarter-sample positions 1..3) and Table 8-12 (chroma, eighth-sample positions 1..7); position 0 = the sample itself (tap 1 at the
// filter's centre position: 3 of 8, 1 of 4), so that one code path serves every case
JM_HD int luma_tap(int f, int i) {
    // (a switch, not a table: usable from host and device alike without a second definition of the array)
    const int t1[8] = {-1, 4, -10, 58, 17, -5, 1, 0}, t2[8] = {-1, 4, -11, 40, 40, -11, 4, -1}, t3[8] = {0, 1, -5, 17, 58, -10, 4, -1};
    return f == 0 ? (i == 3) : (f == 1 ? t1[i] : (f == 2 ? t2[i] : t3[i]));
}
JM_HD int chroma_tap(int f, int i) {
    const int t[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
    return f == 0 ? (i == 1) : t[f][i];
}
// the horizontal taps as signed bytes: ta = taps 0..3, tb = taps 4..7 (luma); chroma: ta = its four taps
JM_HD void luma_taps_h(int f, uint32_t &ta, uint32_t &tb) {
    ta = tb = 0;
    for (int i = 0; i < 4; i++) { ta |= ((uint32_t)luma_tap(f, i) & 255u) << (8 * i); tb |= ((uint32_t)luma_tap(f, 4 + i) & 255u) << (8 * i); }
}
JM_HD uint32_t chroma_taps_h(int f) {
    uint32_t t = 0;
    for (int i = 0; i < 4; i++) t |= ((uint32_t)chroma_tap(f, i) & 255u) << (8 * i);
    return t;
}
// The vertical taps as 16-bit pairs for a column string that starts at an EVEN row (pairs row 2k | row 2k + 1).  The sum for the rows r .. r + n - 1 with r
// even uses the pairs (t0 t1) (t2 t3) ..; with r odd the string is read from row r - 1 and the pairs are (0 t0) (t1 t2) .. (t7 0).  tp[0..4] luma, tp[0..2] chroma.
JM_HD void luma_taps_v(int f, bool odd, uint32_t *tp) {
    for (int k = 0; k < 5; k++) {
        const int i0 = 2 * k - (odd ? 1 : 0), i1 = i0 + 1;
        const int a = i0 >= 0 && i0 < 8 ? luma_tap(f, i0) : 0, b = i1 >= 0 && i1 < 8 ? luma_tap(f, i1) : 0;
        tp[k] = ((uint32_t)a & 0xffffu) | (uint32_t)b << 16;
    }
}
JM_HD void chroma_taps_v(int f, bool odd, uint32_t *tp) {
    for (int k = 0; k < 3; k++) {
        const int i0 = 2 * k - (odd ? 1 : 0), i1 = i0 + 1;
        const int a = i0 >= 0 && i0 < 4 ? chroma_tap(f, i0) : 0, b = i1 >= 0 && i1 < 4 ? chroma_tap(f, i1) : 0;
        tp[k] = ((uint32_t)a & 0xffffu) | (uint32_t)b << 16;
    }
}

// luma, horizontal: h[k] = sum_i tap[i] * s[k + i], k = 0..3, for the eleven signed bytes s[0..10] at the front of (d0 d1 d2)
JM_HD void hsum8x4(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t ta, uint32_t tb, int *h) {
    const uint32_t w1 = alignbyte(d1, d0, 1), w2 = alignbyte(d1, d0, 2), w3 = alignbyte(d1, d0, 3);
    const uint32_t w5 = alignbyte(d2, d1, 1), w6 = alignbyte(d2, d1, 2), w7 = alignbyte(d2, d1, 3);
    h[0] = sdot4(d0, ta, sdot4(d1, tb, 0));
    h[1] = sdot4(w1, ta, sdot4(w5, tb, 0));
    h[2] = sdot4(w2, ta, sdot4(w6, tb, 0));
    h[3] = sdot4(w3, ta, sdot4(w7, tb, 0));
}
// chroma, horizontal: the interleaved bytes Cb0 Cr0 Cb1 Cr1 .. Cb4 Cr4 at the front of (d0 d1 d2) -> h[0] = Cb sum of pair 0, h[1] = Cr of pair 0, h[2] = Cb of
// pair 1, h[3] = Cr of pair 1 (a pair's sum runs over the pairs p .. p + 3)
JM_HD void hsum4x2uv(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t t, int *h) {
    const uint32_t e0 = alignbyte(d1, d0, 2), e1 = alignbyte(d2, d1, 2);                 // the same string from pair 1 on
    h[0] = sdot4(perm(d1, d0, 0x06040200u), t, 0); h[1] = sdot4(perm(d1, d0, 0x07050301u), t, 0);
    h[2] = sdot4(perm(e1, e0, 0x06040200u), t, 0); h[3] = sdot4(perm(e1, e0, 0x07050301u), t, 0);
}
// two rows of four horizontal sums -> four column pairs (row a | row b << 16)
JM_HD void pack_rows(const int *ha, const int *hb, uint32_t *p) {
    for (int k = 0; k < 4; k++) p[k] = ((uint32_t)ha[k] & 0xffffu) | (uint32_t)hb[k] << 16;
}
// vertical: the sum over one column string (dwords from the pair that holds the first row, or the row before it when that one is odd)
JM_HD int vsum8(const uint32_t *col, const uint32_t *tp) { return sdot2(col[0], tp[0], sdot2(col[1], tp[1], sdot2(col[2], tp[2], sdot2(col[3], tp[3], sdot2(col[4], tp[4], 0))))); }
JM_HD int vsum4(const uint32_t *col, const uint32_t *tp) { return sdot2(col[0], tp[0], sdot2(col[1], tp[1], sdot2(col[2], tp[2], 0))); }

// What the two passes leave is  sum - 128 * (sum of the horizontal taps) * (sum of the vertical taps); 8.5.3.3.3.1 / .2 (8-bit: shift1 0, shift2 6, shift3 6):
//   both fractions 0: predSample = sample << 6;  one of them 0: the one sum, unshifted;  neither: the double sum >> 6.
// (luma: tap sums 64 / 1; chroma: 64 / 1 as well -- its position-0 "filter" is written as tap 1, not 64, see chroma_tap)
JM_HD int pred14(int v, bool xf, bool yf) {
    if (xf && yf) return (v + 128 * 64 * 64) >> 6;
    if (xf || yf) return v + 128 * 64;
    return (v + 128) << 6;
}
// 8.5.3.3.4.2, default weighted prediction of four samples: uni-prediction (a + 32) >> 6, bi-prediction (a + b + 64) >> 7, clipped; one byte each
JM_HD uint32_t weigh_default4(const int *a, const int *b, bool both) {
    int v[4];
    for (int k = 0; k < 4; k++) v[k] = both ? (a[k] + b[k] + 64) >> 7 : (a[k] + 32) >> 6;
    const uint32_t lo = perm((uint32_t)v[1], (uint32_t)v[0], 0x05040100u), hi = perm((uint32_t)v[3], (uint32_t)v[2], 0x05040100u);
    return perm(pk::sat_pk_u8(hi), pk::sat_pk_u8(lo), 0x05040100u);
}

// ---- the two passes of one wave over one block, lane by lane (k_hevc_mc; tests/native/hevc_mc_packed_check.cpp runs the same functions for lane 0..63) ----
constexpr int kMcRowDw = 9;                    // tile row: 8 dwords of window + 1 (an odd stride: pass 1 reads two rows per lane, twelve row pairs at once)
constexpr int kMcTileDw = 24 * kMcRowDw;       // window: up to 23 rows (+ 1: pass 1 works on row pairs)
constexpr int kMcColDw = 13;                   // column string of the intermediates: 12 row pairs, + 1 against bank conflicts between neighbouring columns
// pass 1: lane -> (row pair, output dword q of the row): rows 2 * rp and 2 * rp + 1 of the window, the four intermediate columns 4 * q .. 4 * q + 3.
// tile: samples ^ 0x80, rows of kMcRowDw dwords, the window's first sample at byte `sh` of a row; qw: output dwords per row; th: window rows.
// CHROMA: the row is interleaved Cb Cr and the four columns are Cb Cr Cb Cr of two pairs (ta = the four taps, tb unused)
// lane / qw for lane < 64 and qw = 1..4 (output dwords per row: blocks are 4, 8, 12 or 16 samples wide) without an integer division (some thirty instructions)
JM_HD int div_qw(int lane, int qw) { return (lane * (qw == 1 ? 256 : (qw == 2 ? 128 : (qw == 3 ? 86 : 64)))) >> 8; }
template <bool CHROMA> JM_HD void mc_pass1(const uint32_t *tile, int sh, int qw, int th, int lane, uint32_t ta, uint32_t tb, uint32_t *hcol) {
    const int rp = div_qw(lane, qw), q = lane - rp * qw;
    if (rp * 2 >= th) return;
    const uint32_t *r0 = tile + (2 * rp) * kMcRowDw + ((sh + 4 * q) >> 2), *r1 = r0 + kMcRowDw;
    const uint32_t s = (uint32_t)(sh + 4 * q) & 3u;
    uint32_t a[3], b[3];
    for (int i = 0; i < 3; i++) { a[i] = alignbyte(r0[i + 1], r0[i], s); b[i] = alignbyte(r1[i + 1], r1[i], s); }
    int ha[4], hb[4]; uint32_t pr[4];
    if (CHROMA) { hsum4x2uv(a[0], a[1], a[2], ta, ha); hsum4x2uv(b[0], b[1], b[2], ta, hb); }
    else { hsum8x4(a[0], a[1], a[2], ta, tb, ha); hsum8x4(b[0], b[1], b[2], ta, tb, hb); }
    pack_rows(ha, hb, pr);
    for (int i = 0; i < 4; i++) hcol[(4 * q + i) * kMcColDw + rp] = pr[i];
}
// pass 2: the four 14-bit intermediates of output row `row`, output dword `q`; tp: the vertical taps for the row's parity (5 pairs luma, 3 chroma)
template <bool CHROMA> JM_HD void mc_pass2(const uint32_t *hcol, int row, int q, const uint32_t *tp, bool xf, bool yf, int *out) {
    for (int i = 0; i < 4; i++) {
        const uint32_t *col = hcol + (4 * q + i) * kMcColDw + (row >> 1);
        uint32_t c[5];
        for (int k = 0; k < (CHROMA ? 3 : 5); k++) c[k] = col[k];
        out[i] = pred14(CHROMA ? vsum4(c, tp) : vsum8(c, tp), xf, yf);
    }
}

}  // namespace hpk
}  // namespace jmamd
