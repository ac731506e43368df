// jmcodec_amd/csrc/hevc_resid_packed.h -- the inverse transform of HEVC (ITU-T H.265 8.6.4.2) on packed 16-bit pairs (round 6).
//
// What it replaces: rounds 1-5 ran both stages of the separable transform as plain integer code on one 64-thread workgroup per transform block -- per
// multiply-add one byte read of the matrix and one 16-bit read of the data from LDS, the block's matrix copied into LDS for every block, divisions by run-time
// widths in the index arithmetic: 664 VALU + 386 SALU wave-instructions per transform block, 32 M + 18.6 M per 4K picture, the largest bill of any kernel
// in the repository (profiles/r05_sq_counters_c3.json).  Here:
//   * the matrices of all four sizes (and the 4x4 DST) sit in LDS once per workgroup as 16-bit PAIRS along the summation index: mp[jp][y] =
//     (M[2 jp][y], M[2 jp + 1][y]); both stages sum over the first index of M, so one table serves both;
//   * first stage (8.6.4.2 step 1-2, down the columns): the coefficients are scattered into row PAIRS dp[jp][x] = (d[2 jp][x], d[2 jp + 1][x]);
//     g[y][x] = Clip3(-32768, 32767, (sum_jp dot2(mp[jp][y], dp[jp][x]) + 64) >> 7): one v_dot2_i32_i16 per two multiply-adds, only for the row pairs and
//     columns that hold coefficients;
//   * second stage (along the rows): g is stored row-major as 16-bit values, which makes (g[y][2 kp], g[y][2 kp + 1]) a dword:
//     r[y][x] = (sum_kp dot2(mp[kp][x], gdw[y][kp]) + 2048) >> 12; a lane computes two neighbouring x and stores them as one dword;
//   * every index is a shift or a mask (block sizes and the padded column count are powers of two).
// Every function is __host__ __device__ and written per task (one output element or pair): the kernels loop `for (t = lane; t < tasks; t += 64)`, and
// tests/native/hevc_resid_packed_check.cpp runs the same functions task by task against a literal restatement of the clause (tests/test_hevc_resid_packed.py).
//
// Part of the replacement for cuvidDecodePicture with codec_type 1 (/root/reference/nv_dec/nv_dec.cpp:33-41).
#pragma once
#include "hevc_mc_packed.h"

namespace jmamd {
namespace hrp {

using hpk::sdot2;

// the pair tables in one array of dwords: 4-point DCT, 4-point DST, 8-, 16-, 32-point DCT
constexpr int kOff4 = 0, kOffDst = 8, kOff8 = 16, kOff16 = 48, kOff32 = 176, kPairDw = 688;
JM_HD int pair_off(int log2, bool dst) { return dst ? kOffDst : (log2 == 2 ? kOff4 : (log2 == 3 ? kOff8 : (log2 == 4 ? kOff16 : kOff32))); }
// host: trans = the 32-point matrix transMatrix of 8.6.4.2 (coefficient index first), dst = the 4-point DST-VII matrix.  The n-point matrix is rows
// 0, 32/n, 2 * 32/n, .. of the 32-point one, first n columns.
inline void build_pair_table(const int8_t trans[32][32], const int8_t dst[4][4], uint32_t *mp) {
    for (int log2 = 2; log2 <= 5; log2++) {
        const int n = 1 << log2, step = 32 >> log2, off = pair_off(log2, false);
        for (int jp = 0; jp < n / 2; jp++) for (int y = 0; y < n; y++)
            mp[off + jp * n + y] = ((uint32_t)(int)trans[2 * jp * step][y] & 0xffffu) | (uint32_t)(int)trans[(2 * jp + 1) * step][y] << 16;
    }
    for (int jp = 0; jp < 2; jp++) for (int y = 0; y < 4; y++)
        mp[kOffDst + jp * 4 + y] = ((uint32_t)(int)dst[2 * jp][y] & 0xffffu) | (uint32_t)(int)dst[2 * jp + 1][y] << 16;
}

// smallest power of two >= v (v in 1..32), at least 4: the number of columns the first stage computes
JM_HD int pad_cols(int xw) { return xw <= 4 ? 4 : (xw <= 8 ? 8 : (xw <= 16 ? 16 : 32)); }
JM_HD int log2_of(int p) { return p == 4 ? 2 : (p == 8 ? 3 : (p == 16 ? 4 : 5)); }

// where coefficient (row j, column x) of an n x n block goes in the row-pair array, in 16-bit units
JM_HD int pair_slot(int j, int x, int log2) { return ((((j >> 1) << log2) + x) << 1) + (j & 1); }

// first stage, task t of n * cw: column x = t & (cw - 1), output row y = t >> log2(cw); pairs 0 .. jpmax of the coefficients.  Returns g[y][x] and where it goes.
JM_HD int col_task(int t, int log2, int lcw, int jpmax, const uint32_t *mp_n, const uint32_t *dp, int &g_index) {
    const int n = 1 << log2, x = t & ((1 << lcw) - 1), y = t >> lcw;
    int acc = 64;
    for (int jp = 0; jp <= jpmax; jp++) acc = sdot2(mp_n[(jp << log2) + y], dp[(jp << log2) + x], acc);
    acc >>= 7;
    g_index = y * n + x;
    return acc < -32768 ? -32768 : (acc > 32767 ? 32767 : acc);
}
// second stage, task t of n * n / 2: output row y = t >> (log2 - 1), columns x = 2 * (t & (n / 2 - 1)) and x + 1; kpn = (padded columns) / 2 pairs of g.
// gdw: g as dwords (n / 2 per row).  Returns the two residuals as one dword (x | x + 1 << 16), to be stored at dword index t.
JM_HD uint32_t row_task(int t, int log2, int kpn, const uint32_t *mp_n, const uint32_t *gdw) {
    const int hn = 1 << (log2 - 1), y = t >> (log2 - 1), x = (t & (hn - 1)) << 1;
    int a0 = 2048, a1 = 2048;
    for (int kp = 0; kp < kpn; kp++) {
        const uint32_t gd = gdw[y * hn + kp];
        a0 = sdot2(mp_n[(kp << log2) + x], gd, a0); a1 = sdot2(mp_n[(kp << log2) + x + 1], gd, a1);
    }
    return ((uint32_t)(a0 >> 12) & 0xffffu) | (uint32_t)(a1 >> 12) << 16;
}
// transform skip (8.6.4.2, rotate-free form of this profile): r = ((d << 7) + 2048) >> 12; transquant bypass: r = d
JM_HD int tskip_value(int d) { return ((d << 7) + 2048) >> 12; }

}  // namespace hrp
}  // namespace jmamd
