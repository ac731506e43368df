// jmcodec_amd/csrc/hevc_kernels.hip -- HEVC sample reconstruction on gfx950 (ITU-T H.265 8.4.4.2, 8.5.3.3, 8.6.4.2, 8.7.2.5, 8.7.3).
//
// The device half of the replacement for cuvidDecodePicture with codec_type 1 (/root/reference/nv_dec/nv_dec.cpp:33-41): the host
// (hevc_slice.cpp) delivers motion-compensation blocks, scaled coefficients, intra blocks with their neighbour availability,
// boundary strengths and SAO parameters (hevc_jobs.h); these kernels produce the samples.  One launch serves a batch of pictures
// (blockIdx.y = picture), as for H.264.
//   k_hevc_mc        one wavefront per block of <= 16x16 luma samples: 8-tap / 4-tap separable interpolation through LDS, weighting
//   k_hevc_resid     one wavefront per transform block of an inter CU: sparse coefficients -> inverse transform -> add
//   k_hevc_intra     one workgroup per CTB row, rows chained by progress counters: the CTB's intra blocks in decoding order, CTB held in LDS
//   k_hevc_deblock   one lane per 4-sample edge segment; all vertical edges of the picture, then all horizontal edges (8.7.2)
//   k_hevc_sao       one lane per dword of a surface row, deblocked surface -> final surface (8.7.3)
// All integer arithmetic on 8-bit samples; surfaces are pitch-linear NV12 like the H.264 path's.
#include <hip/hip_runtime.h>
#include "hevc_jobs.h"
#include "hevc_kernels.h"
#include "hevc_tables.h"
#include "hevc_mc_packed.h"
#include "hevc_resid_packed.h"
#include "hevc_bs.h"
#include "chain_common.h"
#include <cstdlib>

namespace jmamd {

__constant__ int8_t c_lf[4][8];
__constant__ int8_t c_cf[8][4];
__constant__ int8_t c_angle[35];
__constant__ int16_t c_inv_angle[35];
__constant__ uint8_t c_beta[52], c_tc[54], c_qpc[58];

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int clip1(int v) { return clip3(0, 255, v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
// sample address in an NV12 surface: plane 0 luma, 1 / 2 the interleaved chroma components
__device__ __forceinline__ uint8_t *sample_ptr(uint8_t *surf, const HevcPicParams &pp, int c, int x, int y) {
    return c == 0 ? surf + (size_t)y * pp.pitch + x : surf + pp.chroma_offset + (size_t)y * pp.pitch + 2 * x + (c - 1);
}

// ------------------------------------------------------------------------------------------------------------
// 8.5.3.3: motion compensation
// ------------------------------------------------------------------------------------------------------------
// One WAVEFRONT per prediction block of at most 16x16 luma samples, four of them to a workgroup (round 6; until round 5 one 64-thread workgroup per block
// and plain integer filters: 29.7 M VALU wave-instructions per 4K picture).  The sample arithmetic is hevc_mc_packed.h.  Per reference list:
//   window  the (w+7) x (h+7) luma window (then the interleaved CbCr window, (w/2+3) x (h/2+3) pairs) goes into the wave's LDS tile as ALIGNED DWORDS of
//           32-byte rows, samples XOR 0x80 -- unless it touches the picture border (then byte-wise with clamped coordinates, 8.5.3.3.3.1);
//   pass 1  a lane takes two window rows and four columns: horizontal sums by v_dot4, packed into 16-bit row pairs, stored column-major (one round);
//   pass 2  a lane takes one row and four columns (luma) / two CbCr pairs (chroma) of the block: five (three) v_dot2 per sample down the column strings;
// every lane ends up with four 14-bit intermediates and stores ONE dword (block positions and widths are multiples of 4).  Cb and Cr share one window load
// and one store.  The wave's LDS operations execute in order, so a compiler barrier stands where a workgroup would need s_barrier.
using hpk::kMcRowDw; using hpk::kMcTileDw; using hpk::kMcColDw;
struct HevcMcLds {
    uint32_t tile[kMcTileDw];
    uint32_t hcol[16 * kMcColDw];              // [column][row pair]
};
__constant__ uint32_t c_lh[4][2];              // hpk::luma_taps_h
__constant__ uint32_t c_lv[4][2][5];           // hpk::luma_taps_v [fraction][row parity]
__constant__ uint32_t c_ch[8];                 // hpk::chroma_taps_h
__constant__ uint32_t c_cv[8][2][3];           // hpk::chroma_taps_v

// the window of one list into the tile.  row_bytes: bytes of a window row (luma tw, chroma 2 * tw); x0b: byte column of its first sample in the plane row;
// inside: it lies within the plane (and the last dword of a row within the pitch).  Returns the byte offset of the first sample in a tile row.
__device__ __forceinline__ int mc_fill_tile(const uint8_t *plane, int pitch, int x0b, int y0, int row_bytes, int th, bool inside, int pw_b, int ph, int step,
    uint32_t *tile, int lane) {
    if (inside) {
        const int sh = x0b & 3, ndw = (sh + row_bytes + 3) >> 2;          // <= 7
        const uint8_t *base = plane + (size_t)y0 * pitch + (x0b & ~3);
        for (int k = lane; k < th * 8; k += 64) { const int r = k >> 3, d = k & 7;
            if (d < ndw) tile[r * kMcRowDw + d] = *(const JM_GLOBAL_AS uint32_t *)(base + (size_t)r * pitch + 4 * d) ^ 0x80808080u; }
        return sh;
    }
    // border: byte-wise, coordinates clamped per sample (chroma: per CbCr pair, step 2)
    for (int k = lane; k < th * 8; k += 64) {
        const int r = k >> 3, d = k & 7;
        if (4 * d >= row_bytes) continue;
        const JM_GLOBAL_AS uint8_t *row = (const JM_GLOBAL_AS uint8_t *)plane + (size_t)clip3(0, ph - 1, y0 + r) * pitch;
        uint32_t w = 0;
        for (int i = 0; i < 4; i++) { const int xb = x0b + 4 * d + i; const int xs = clip3(0, pw_b - step, xb & ~(step - 1)) + (xb & (step - 1));
            w |= (uint32_t)row[xs] << (8 * i); }
        tile[r * kMcRowDw + d] = w ^ 0x80808080u;
    }
    return 0;
}

// luma: this lane's four 14-bit intermediates (row my_row, columns 4 * my_q ..) of one list's prediction
__device__ __forceinline__ void hevc_mc_luma(const HevcPicParams &pp, const uint8_t *ref, int xi, int yi, int bw, int bh, int xf, int yf, HevcMcLds &sm,
    int lane, bool mine, int my_row, int my_q, int *out) {
    const int tw = bw + 7, th = bh + 7, x0 = xi - 3, y0 = yi - 3, qw = bw >> 2;
    const bool inside = x0 >= 0 && y0 >= 0 && x0 + tw <= pp.w && y0 + th <= pp.h && ((x0 + tw + 3) & ~3) <= pp.pitch;
    __builtin_amdgcn_wave_barrier();                              // (the previous user of the tile is done: same wave, in order)
    const int sh = mc_fill_tile(ref, pp.pitch, x0, y0, tw, th, inside, pp.w, pp.h, 1, sm.tile, lane);
    __builtin_amdgcn_wave_barrier();
    hpk::mc_pass1<false>(sm.tile, sh, qw, th, lane, c_lh[xf][0], c_lh[xf][1], sm.hcol);       // lane -> (row pair, column quad); at most 12 x 4 tasks
    __builtin_amdgcn_wave_barrier();
    if (!mine) return;
    uint32_t tp[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { const uint32_t te = c_lv[yf][0][k], to = c_lv[yf][1][k]; tp[k] = (my_row & 1) ? to : te; }      // (scalar loads + a select)
    hpk::mc_pass2<false>(sm.hcol, my_row, my_q, tp, xf != 0, yf != 0, out);                   // rows my_row .. my_row + 7 of the intermediates, down four columns
}
// chroma: this lane's two CbCr pairs (Cb0, Cr0, Cb1, Cr1 of row my_row, pair columns 2 * my_q, 2 * my_q + 1)
__device__ __forceinline__ void hevc_mc_chroma(const HevcPicParams &pp, const uint8_t *refc, int xi, int yi, int bw, int bh, int xf, int yf, HevcMcLds &sm,
    int lane, bool mine, int my_row, int my_q, int *out) {
    const int tw = bw + 3, th = bh + 3, x0 = xi - 1, y0 = yi - 1, pw = pp.w >> 1, ph = pp.h >> 1, qw = bw >> 1;
    const bool inside = x0 >= 0 && y0 >= 0 && x0 + tw <= pw && y0 + th <= ph && ((2 * (x0 + tw) + 3) & ~3) <= pp.pitch;
    __builtin_amdgcn_wave_barrier();
    const int sh = mc_fill_tile(refc, pp.pitch, 2 * x0, y0, 2 * tw, th, inside, 2 * pw, ph, 2, sm.tile, lane);
    __builtin_amdgcn_wave_barrier();
    // pass 1: lane -> (row pair, two CbCr pairs); the intermediates' "columns" are the bytes of the interleaved output row: Cb0 Cr0 Cb1 Cr1 ..
    hpk::mc_pass1<true>(sm.tile, sh, qw, th, lane, c_ch[xf], 0u, sm.hcol);
    __builtin_amdgcn_wave_barrier();
    if (!mine) return;
    uint32_t tp[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { const uint32_t te = c_cv[yf][0][k], to = c_cv[yf][1][k]; tp[k] = (my_row & 1) ? to : te; }
    hpk::mc_pass2<true>(sm.hcol, my_row, my_q, tp, xf != 0, yf != 0, out);                    // out[i]: byte i of the output dword = Cb0 Cr0 Cb1 Cr1
}

constexpr int kMcWaves = 4;
__global__ __launch_bounds__(64 * kMcWaves) void k_hevc_mc(const HevcPicParams *pics) {
    const HevcPicParams &pp = pics[blockIdx.y];
    // XCD-aware: workgroups are dealt round-robin to the 8 XCDs (each with its own L2); give every XCD one contiguous run of blocks
    // (blocks are in decoding order, i.e. spatially coherent), so that a reference window is fetched into one L2 instead of eight
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int per_xcd = ((int)gridDim.x + 7) >> 3, job = (((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3)) * kMcWaves + wave;
    if (!(pp.stages & HPS_MC) || job >= pp.n_pus) return;      // (no workgroup barrier below: a wave may leave alone)
    const HevcPu pu = pp.pus[job];
    const int lane = (int)(threadIdx.x & 63);
    __shared__ HevcMcLds sm_all[kMcWaves];
    HevcMcLds &sm = sm_all[wave];
    uint8_t *dst = pp.work_surf;
    const HevcWp *wp = pu.wp ? &pp.wps[pu.wp - 1] : nullptr;
    const bool both = pu.slot0 >= 0 && pu.slot1 >= 0;
    // 8.5.3.3.4.3 (explicit weights): the two 14-bit predictions -> one sample of component c.  Default weights: hpk::weigh_default4
    auto weigh = [&](int p0, int p1, int c) -> int {
        int v;
        const int ld = wp->log2wd[c ? 1 : 0];
        if (both) v = (p0 * wp->w[0][pu.ridx0][c] + p1 * wp->w[1][pu.ridx1][c] + ((wp->o[0][pu.ridx0][c] + wp->o[1][pu.ridx1][c] + 1) << ld)) >> (ld + 1);
        else if (pu.slot0 >= 0) v = ((p0 * wp->w[0][pu.ridx0][c] + (1 << (ld - 1))) >> ld) + wp->o[0][pu.ridx0][c];
        else v = ((p1 * wp->w[1][pu.ridx1][c] + (1 << (ld - 1))) >> ld) + wp->o[1][pu.ridx1][c];
        return clip1(v);
    };
    {   // ---- luma: lane -> (row, dword) ----
        const int bw = pu.w, bh = pu.h, qw = bw >> 2, my_row = hpk::div_qw(lane, qw), my_q = lane - my_row * qw;
        const bool mine = lane < bh * qw;
        int p[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int l = 0; l < 2; l++) {
            const int slot = l ? pu.slot1 : pu.slot0;
            if (slot < 0) continue;
            const int mvx = l ? pu.mv1[0] : pu.mv0[0], mvy = l ? pu.mv1[1] : pu.mv0[1];
            hevc_mc_luma(pp, pp.surf[slot], pu.x + (mvx >> 2), pu.y + (mvy >> 2), bw, bh, mvx & 3, mvy & 3, sm, lane, mine, my_row, my_q, p[l]);
        }
        if (mine) {
            uint32_t w = 0;
            if (!wp) w = both ? hpk::weigh_default4(p[0], p[1], true) : hpk::weigh_default4(pu.slot0 >= 0 ? p[0] : p[1], p[1], false);
            else for (int i = 0; i < 4; i++) w |= (uint32_t)weigh(p[0][i], p[1][i], 0) << (8 * i);
            *(JM_GLOBAL_AS uint32_t *)(dst + (size_t)(pu.y + my_row) * pp.pitch + pu.x + 4 * my_q) = w;
        }
    }
    {   // ---- chroma, both components: lane -> (row, dword of two CbCr pairs) ----
        const int bw = pu.w >> 1, bh = pu.h >> 1, qw = bw >> 1, my_row = hpk::div_qw(lane, qw), my_q = lane - my_row * qw;
        const bool mine = lane < bh * qw;
        int p[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int l = 0; l < 2; l++) {
            const int slot = l ? pu.slot1 : pu.slot0;
            if (slot < 0) continue;
            const int mvx = l ? pu.mv1[0] : pu.mv0[0], mvy = l ? pu.mv1[1] : pu.mv0[1];
            hevc_mc_chroma(pp, pp.surf[slot] + pp.chroma_offset, (pu.x >> 1) + (mvx >> 3), (pu.y >> 1) + (mvy >> 3), bw, bh, mvx & 7, mvy & 7, sm, lane, mine,
                my_row, my_q, p[l]);
        }
        if (mine) {
            uint32_t w = 0;
            if (!wp) w = both ? hpk::weigh_default4(p[0], p[1], true) : hpk::weigh_default4(pu.slot0 >= 0 ? p[0] : p[1], p[1], false);
            else for (int i = 0; i < 4; i++) w |= (uint32_t)weigh(p[0][i], p[1][i], 1 + (i & 1)) << (8 * i);
            *(JM_GLOBAL_AS uint32_t *)(dst + pp.chroma_offset + (size_t)((pu.y >> 1) + my_row) * pp.pitch + pu.x + 4 * my_q) = w;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// 8.6.4.2: residual of one transform block from its sparse scaled coefficients
// ------------------------------------------------------------------------------------------------------------
// Round 6 (hevc_resid_packed.h): one WAVE per transform block, four waves to a workgroup, every wave takes kResidPerWave consecutive blocks of the list; the
// matrices of all sizes sit in LDS once per workgroup as 16-bit pairs; both stages are v_dot2_i32_i16 sums over the row pairs / column pairs that hold
// coefficients.  (Until round 5: one 64-thread workgroup per block, the block's matrix copied to LDS per block, byte-wise multiply-adds.)
constexpr int kResidWaves = 4, kResidPerWave = 8, kIresidPerWave = 2;
constexpr int kResidBufDw = 512;                       // one wave's coefficient pairs (16 x 32) / intermediate (32 x 16 dwords)
__device__ uint32_t g_resid_pairs[hrp::kPairDw];       // hrp::build_pair_table, uploaded once per device
struct HevcResidLds {
    uint32_t mp[hrp::kPairDw];
    uint32_t dp[kResidWaves][kResidBufDw];
    uint32_t g[kResidWaves][kResidBufDw];
};
// largest value of v over the wave, wave-uniform (v >= 0)
__device__ __forceinline__ int wave_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));         // quad_perm [1 0 3 2]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));         // quad_perm [2 3 0 1]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));        // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false));        // row_mirror
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// The residual of one block, by ONE wave (its LDS operations execute in order: a compiler barrier stands where a workgroup would need s_barrier).
// Returns the residual as dwords of two neighbouring samples (x | x + 1 << 16), n / 2 per row -- in dp, or in g for transform-skip / bypass blocks.
__device__ __forceinline__ const uint32_t *residual_block(const uint32_t *coefs, int count, int log2, int flags, uint32_t *dp, uint32_t *g, const uint32_t *mp,
    int lane) {
    const int n = 1 << log2, nn = n * n;
    const uint32_t e0 = lane < count ? coefs[lane] : 0u;                          // (most blocks hold fewer than 64 coefficients: one load serves both walks)
    __builtin_amdgcn_wave_barrier();
    if (flags & (HTB_BYPASS | HTB_TSKIP)) {
        for (int k = lane; k < (nn >> 1); k += 64) g[k] = 0;
        __builtin_amdgcn_wave_barrier();
        int16_t *r16 = (int16_t *)g;
        for (int k = lane; k < count; k += 64) { const uint32_t e = k < 64 ? e0 : coefs[k]; const int v = (int16_t)(e >> 16);
            r16[e & 1023u] = (int16_t)((flags & HTB_BYPASS) ? v : hrp::tskip_value(v)); }
        __builtin_amdgcn_wave_barrier();
        return g;
    }
    // which row pairs and columns hold coefficients
    int mj = 0, mx = 0;
    for (int k = lane; k < count; k += 64) { const uint32_t e = k < 64 ? e0 : coefs[k]; const int pos = (int)(e & 1023u);
        mj = max(mj, pos >> log2); mx = max(mx, pos & (n - 1)); }
    const int jpmax = wave_max(mj) >> 1, cw = min(hrp::pad_cols(wave_max(mx) + 1), n), lcw = hrp::log2_of(cw);
    for (int k = lane; k < ((jpmax + 1) << log2); k += 64) dp[k] = 0;
    __builtin_amdgcn_wave_barrier();
    int16_t *d16 = (int16_t *)dp;
    for (int k = lane; k < count; k += 64) { const uint32_t e = k < 64 ? e0 : coefs[k]; const int pos = (int)(e & 1023u);
        d16[hrp::pair_slot(pos >> log2, pos & (n - 1), log2)] = (int16_t)(e >> 16); }
    __builtin_amdgcn_wave_barrier();
    const uint32_t *mp_n = mp + hrp::pair_off(log2, (flags & HTB_DST) != 0);
    int16_t *g16 = (int16_t *)g;
    for (int t = lane; t < (n << lcw); t += 64) { int gi; const int v = hrp::col_task(t, log2, lcw, jpmax, mp_n, dp, gi); g16[gi] = (int16_t)v; }
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < (nn >> 1); t += 64) dp[t] = hrp::row_task(t, log2, cw >> 1, mp_n, g);        // (the second stage reads g only: dp is free again)
    __builtin_amdgcn_wave_barrier();
    return dp;
}

__global__ __launch_bounds__(64 * kResidWaves) void k_hevc_resid(const HevcPicParams *pics) {
    const HevcPicParams &pp = pics[blockIdx.y];
    const int per_xcd = ((int)gridDim.x + 7) >> 3, wg = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);      // XCD-aware, see k_hevc_mc
    if (!(pp.stages & HPS_RESID) || wg * (kResidWaves * kResidPerWave) >= pp.n_tbs) return;
    __shared__ HevcResidLds sm;
    for (int k = threadIdx.x; k < hrp::kPairDw; k += 64 * kResidWaves) sm.mp[k] = g_resid_pairs[k];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    uint32_t *dp = sm.dp[wave], *g = sm.g[wave];
    uint8_t *dst = pp.work_surf;
    // Cb and Cr of a transform unit are interleaved in memory and are two entries of the list, one behind the other when both are coded (transform_unit()
    // emits c = 1, then c = 2).  The Cb wave takes its Cr partner along, so the pair leaves as whole dwords (Cb Cr Cb Cr) instead of byte-wise
    // read-modify-writes of two waves on the same lines; a component whose partner is not coded does the same with a zero residual for the other half
    // (nobody else writes those bytes: blocks are disjoint, and the motion compensation ran in an earlier kernel).
    auto same_place = [](const HevcTb &a, const HevcTb &b) { return a.x == b.x && a.y == b.y && a.log2 == b.log2; };
    const int first = (wg * kResidWaves + wave) * kResidPerWave;
    for (int job = first; job < first + kResidPerWave && job < pp.n_tbs; job++) {
        const HevcTb tb = pp.tbs[job];
        if (!tb.coef_n) continue;                                     // (a luma block whose levels all scaled to zero: listed for the boundary strengths only)
        if (tb.plane == 2 && job > 0) { const HevcTb prev = pp.tbs[job - 1]; if (prev.plane == 1 && same_place(prev, tb)) continue; }   // done by its Cb partner
        const int n = 1 << tb.log2, hn = n >> 1;
        const uint32_t *r = residual_block(pp.coefs + tb.coef_off, (int)tb.coef_n, tb.log2, tb.flags, dp, g, sm.mp, lane);
        if (tb.plane == 0) {
            // luma: four samples of a row per lane, one dword read and one dword written (block positions and sizes are multiples of 4)
            for (int k = lane; k < (n << (tb.log2 - 2)); k += 64) {
                const int y = k >> (tb.log2 - 2), xq = k & ((n >> 2) - 1);
                uint32_t *p = (uint32_t *)(dst + (size_t)(tb.y + y) * pp.pitch + tb.x + 4 * xq);
                *p = pk::add_residual4(*p, r[y * hn + 2 * xq], r[y * hn + 2 * xq + 1]);
            }
            continue;
        }
        // chroma (blocks of at most 16x16: the upper halves of the wave's buffers hold the partner's block)
        const uint32_t *rb = tb.plane == 1 ? r : nullptr, *rr2 = tb.plane == 2 ? r : nullptr;
        if (tb.plane == 1 && job + 1 < pp.n_tbs) {
            const HevcTb nx = pp.tbs[job + 1];
            if (nx.plane == 2 && same_place(tb, nx))
                rr2 = residual_block(pp.coefs + nx.coef_off, (int)nx.coef_n, nx.log2, nx.flags, dp + kResidBufDw / 2, g + kResidBufDw / 2, sm.mp, lane);
        }
        for (int k = lane; k < (n << (tb.log2 - 1)); k += 64) {       // two sample pairs per lane: Cb0 Cr0 Cb1 Cr1
            const int y = k >> (tb.log2 - 1), xq = k & (hn - 1);
            uint32_t *p = (uint32_t *)(dst + pp.chroma_offset + (size_t)(tb.y + y) * pp.pitch + 2 * (tb.x + 2 * xq));
            const uint32_t b = rb ? rb[y * hn + xq] : 0u, c = rr2 ? rr2[y * hn + xq] : 0u;
            *p = pk::add_residual4(*p, pk::perm(c, b, 0x05040100u), pk::perm(c, b, 0x07060302u));
        }
    }
}

// residual of the intra blocks, computed ahead of the (sequential) intra pass: it does not depend on the prediction.  Planar int16
// scratch: Y (w x h), Cb, Cr.
__global__ __launch_bounds__(64 * kResidWaves) void k_hevc_iresid(const HevcPicParams *pics) {
    const HevcPicParams &pp = pics[blockIdx.y];
    // (kIresidPerWave: an I picture's blocks are few and large -- 10 k per 4K picture -- so fewer per wave than in k_hevc_resid keeps the machine covered)
    if (!(pp.stages & HPS_INTRA) || (int)blockIdx.x * (kResidWaves * kIresidPerWave) >= pp.n_itbs) return;
    __shared__ HevcResidLds sm;
    for (int k = threadIdx.x; k < hrp::kPairDw; k += 64 * kResidWaves) sm.mp[k] = g_resid_pairs[k];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    const int first = ((int)blockIdx.x * kResidWaves + wave) * kIresidPerWave;
    for (int job = first; job < first + kIresidPerWave && job < pp.n_itbs; job++) {
        const HevcIntraTb tb = pp.itbs[job];
        if (!tb.coef_n) continue;
        const int n = 1 << tb.log2, hn = n >> 1;
        const uint32_t *r = residual_block(pp.coefs + tb.coef_off, (int)tb.coef_n, tb.log2, tb.flags, sm.dp[wave], sm.g[wave], sm.mp, lane);
        const int pw = tb.plane ? pp.w >> 1 : pp.w;
        int16_t *dst = pp.resid + (tb.plane == 0 ? 0 : (size_t)pp.w * pp.h + (tb.plane == 2 ? (size_t)(pp.w >> 1) * (pp.h >> 1) : 0));
        // (block positions are multiples of 4 and plane widths even: a pair of samples is an aligned dword of the scratch)
        for (int k = lane; k < (n << (tb.log2 - 1)); k += 64) { const int y = k >> (tb.log2 - 1), xp = k & (hn - 1);
            *(uint32_t *)(dst + (size_t)(tb.y + y) * pw + tb.x + 2 * xp) = r[k]; }
    }
}

// sum of v over the 64 lanes of the wave, as a wave-uniform value: four DPP adds give every lane the total of its row of 16, four v_readlane add the rows
__device__ __forceinline__ int wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);          // quad_perm [1 0 3 2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);          // quad_perm [2 3 0 1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);         // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);         // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// ------------------------------------------------------------------------------------------------------------
// 8.4.4.2: intra prediction (+ residual) of the intra blocks of one coding tree block, in decoding order
// ------------------------------------------------------------------------------------------------------------
constexpr int kIntraThreads = 256;
constexpr int kYS = 160, kCS = 80;              // LDS row strides of the luma / chroma tiles
// column of the CTB's first sample inside a tile row (the left neighbour column sits just before): keeps 16- / 8-byte accesses aligned
constexpr int kYO = 16, kCO = 8;
constexpr int kIntraMaxRun = (8192 / 16 + kHevcIntraSegs - 1) / kHevcIntraSegs;      // CTBs of a run: a CTB row of the widest picture / segments
constexpr int kIntraMaxTbs = 448;               // 64x64 CTB: <= 256 + 128 4x4 blocks (+ PCM planes)

// The coding tree block lives in LDS while its intra blocks are reconstructed: tile row 0 / column 0 hold the samples above / left
// of the CTB (row 0 continues over the CTB to the right: above-right neighbours), so every neighbour a block may use (8.4.4.2.2) is
// an LDS read, whether it was produced by this CTB's earlier blocks, by the inter kernels or by a CTB of an earlier diagonal.
//
// One workgroup per CTB row walks the CTBs of its row THAT HOLD INTRA BLOCKS, left to right (the others were final before the launch);
// CTB (x, y) may start when the CTBs above it that hold intra blocks -- left, straight, right -- are done.  Progress counters in device
// memory carry that dependency between workgroups: row y only ever waits for row y - 1, whose workgroup has the smaller block index and
// is therefore already running or finished -- no deadlock whatever the occupancy.  (A launch per wavefront diagonal, the first version,
// cost the sum of the slowest CTB of every diagonal: 23 ms for a batch of 4K pictures; this form follows the real dependency chain.)
//
// What lies ON that chain (round 3; it was everything: 9-10 us per CTB, 550 us for a P picture with scattered intra blocks): the wait for the
// row above, the load of the row of samples above the CTB, the block loop, the store of the CTB's BOTTOM row and the release of the
// counter.  Off the chain: the CTB's own samples (they were final before the launch: fetched into registers while the previous CTB is
// worked on), its left column (out of the previous tile when that was the neighbour), and the store of the rest of the tile (after the
// counter is released; the next CTB's tile is filled from registers, so the stores drain meanwhile).
__global__ __launch_bounds__(kIntraThreads) void k_hevc_intra(const HevcPicParams *pics, int *progress, int prog_stride) {
    // The picture's parameters through a CONSTANT-address-space reference: scalar loads the compiler may hoist and reuse.  Through the generic reference of
    // rounds 1-4 every `pp.x` after a store was a vector load from memory followed by s_waitcnt vmcnt(0) -- one of them (`pp.strong_intra`) in the filter
    // decision of every block, i.e. a memory round trip per block on the picture's dependency chain (round 5, found with wall-clock probes in the kernel).
    typedef const __attribute__((address_space(4))) HevcPicParams ConstPic;
    ConstPic &pp = *(ConstPic *)(uintptr_t)(pics + blockIdx.y);
    if (!(pp.stages & HPS_INTRA)) return;
    // Each CTB row is cut into kHevcIntraSegs runs of CTBs, one workgroup each (block index = row * segments + segment: whatever a workgroup waits
    // for -- CTBs of the row above, the last CTB of the run to its left -- has a smaller block index).  A run publishes "my CTBs up to column c are done"
    // in its own counter.  Where intra blocks are scattered (P / B pictures) few CTBs wait for anything, and the runs of a row work side by side.
    const int cy = (int)blockIdx.x / kHevcIntraSegs, seg = (int)blockIdx.x % kHevcIntraSegs;
    if (cy >= pp.ctb_h) return;
    const int seg_w = (pp.ctb_w + kHevcIntraSegs - 1) / kHevcIntraSegs, c0 = seg * seg_w, c1 = min(pp.ctb_w, c0 + seg_w);
    if (c0 >= c1) return;
    int *prog = progress + (size_t)blockIdx.y * prog_stride;
    __shared__ __align__(16) uint8_t ty[65 * kYS];            // luma tile
    __shared__ __align__(16) uint8_t tc[2][33 * kCS];         // Cb, Cr tiles
    // per colour plane (= per wavefront of the block loop):
    __shared__ int16_t edge_[3][2][132];        // [0] raw, [1] filtered: 0 .. 2n-1 left column bottom-to-top, 2n corner, 2n+1 .. 4n top row left-to-right
    __shared__ int16_t refa_[3][32 * 3 + 8];    // main reference of the angular modes, index 0 at refa[32]
    __shared__ HevcIntraTb s_tbs[kIntraMaxTbs];  // the CTB's block records, fetched once: the block loop touches no global memory at all (round 5)
    // Round 5: the CTB's residual (k_hevc_iresid's planar int16 scratch) travels with the CTB's samples -- fetched into registers while the PREVIOUS CTB is
    // worked on, put into LDS when this one starts.  Before, every block began with its own residual loads and ended waiting for them: one memory round
    // trip per block on the dependency chain of the picture (~19 luma blocks per 64x64 CTB of an I picture).
    __shared__ int16_t s_angle[35], s_inv_angle[35];   // intraPredAngle / invAngle by mode (Tables 8-4 / 8-5): LDS copies -- the __constant__ arrays are reached
                                                       // with vector loads from memory, another round trip per angular block
    __shared__ HevcCtb s_ctb[kIntraMaxRun];
    __shared__ uint8_t s_edge_above[kIntraMaxRun + 4];
    __shared__ uint16_t s_list[3][kIntraMaxTbs];   // per colour plane: the indices of its blocks in s_tbs, in decoding order
    __shared__ __align__(16) int16_t tr_y[64 * 64];
    __shared__ __align__(16) int16_t tr_c[2][32 * 32];
    uint8_t *surf = pp.work_surf;
    uint8_t *cpl = surf + pp.chroma_offset;
    const int cs = 1 << pp.ctb_log2, y0 = cy << pp.ctb_log2, hc = cs >> 1, yc0 = y0 >> 1, pw = pp.w >> 1, ph = pp.h >> 1;
    const int q = cs >> 4;                                     // 16-byte groups per luma row of the CTB (and per interleaved chroma row)
    const int tid = threadIdx.x;
    // this thread's share of a CTB body: luma group (row yr, group yg), chroma group (row cr, group cg) -- at most one each (64x64: 256 + 128 groups)
    const int yr = tid / q, yg = tid - yr * q, cr = tid / q, cg = tid - cr * q;
    const bool y_mine = tid < cs * q, c_mine = tid < hc * q;
    // tb[]: the first 3 x 256 dwords of the CTB's block records (153 of the 20-byte records; CTBs with more fetch the rest when they start)
    struct Pre { uint4 y, c; uint32_t ly, lc; uint4 ry[2]; uint2 rc[2]; uint32_t tb[3]; };
    auto de_interleave = [](const uint4 v, uint2 &cb, uint2 &crv) {
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
        uint32_t b2[2], r2[2];
        for (int h2 = 0; h2 < 2; h2++) { const uint32_t a = w4[2 * h2], b = w4[2 * h2 + 1];
            b2[h2] = (a & 255) | ((a >> 16 & 255) << 8) | ((b & 255) << 16) | ((b >> 16 & 255) << 24);
            r2[h2] = (a >> 8 & 255) | ((a >> 24) << 8) | ((b >> 8 & 255) << 16) | ((b >> 24) << 24); }
        cb = make_uint2(b2[0], b2[1]); crv = make_uint2(r2[0], r2[1]);
    };
    auto interleave = [](const uint2 cb, const uint2 crv) {
        const uint32_t bw[2] = {cb.x, cb.y}, rw[2] = {crv.x, crv.y};
        uint32_t o[4];
        for (int h2 = 0; h2 < 2; h2++) {
            o[2 * h2] = (bw[h2] & 255) | ((rw[h2] & 255) << 8) | ((bw[h2] >> 8 & 255) << 16) | ((rw[h2] >> 8 & 255) << 24);
            o[2 * h2 + 1] = (bw[h2] >> 16 & 255) | ((rw[h2] >> 16 & 255) << 8) | ((bw[h2] >> 24) << 16) | ((rw[h2] >> 24) << 24);
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    };
    // the CTB's own samples and its left column, as they are in memory now (final for everything this kernel does not write)
    auto prefetch = [&](int cxn, Pre &pre) {
        const int x0 = cxn << pp.ctb_log2, xc0 = x0 >> 1;
        pre.y = pre.c = make_uint4(0, 0, 0, 0); pre.ly = pre.lc = 0;
        if (y_mine && y0 + yr < pp.h && x0 + 16 * yg < pp.w) pre.y = *(const uint4 *)(surf + (size_t)(y0 + yr) * pp.pitch + x0 + 16 * yg);
        if (c_mine && yc0 + cr < ph && xc0 + 8 * cg < pw) pre.c = *(const uint4 *)(cpl + (size_t)(yc0 + cr) * pp.pitch + 2 * (xc0 + 8 * cg));
        if (x0 > 0 && tid <= cs) { const int y = y0 + tid - 1; if (y >= 0 && y < pp.h) pre.ly = surf[(size_t)y * pp.pitch + x0 - 1]; }
        if (x0 > 0 && tid <= hc) { const int y = yc0 + tid - 1; if (y >= 0 && y < ph) pre.lc = *(const uint16_t *)(cpl + (size_t)y * pp.pitch + 2 * (xc0 - 1));
            }
        // residual: luma 64 rows x 8 pieces of 8 int16 (two pieces per thread), chroma 2 planes x 32 rows x 4 pieces (one per thread, as two 8-byte loads:
        // a chroma row starts at a multiple of 8 bytes only)
        for (int t = 0; t < 2; t++) { const int pc = tid + 256 * t, row = pc >> 3, sg = pc & 7;
            pre.ry[t] = make_uint4(0, 0, 0, 0);
            if (row < cs && 8 * sg < cs && y0 + row < pp.h && x0 + 8 * sg < pp.w) pre.ry[t] = *(const uint4 *)(pp.resid + (size_t)(y0 + row) * pp.w + x0 + 8 * sg); }
        { const int pl = tid >> 7, row = (tid & 127) >> 2, sg = tid & 3;
            pre.rc[0] = pre.rc[1] = make_uint2(0, 0);
            if (row < hc && 8 * sg < hc && yc0 + row < ph && xc0 + 8 * sg < pw) {
                const int16_t *src = pp.resid + (size_t)pp.w * pp.h + (pl ? (size_t)pw * ph : 0) + (size_t)(yc0 + row) * pw + xc0 + 8 * sg;
                pre.rc[0] = *(const uint2 *)src; pre.rc[1] = *(const uint2 *)(src + 4); } }
        { const HevcCtb &cn = s_ctb[cxn - c0]; const int nd = (int)cn.intra_count * 5; const uint32_t *src = (const uint32_t *)(pp.itbs + cn.intra_first);
            for (int t = 0; t < 3; t++) { const int k = tid + kIntraThreads * t; pre.tb[t] = (cn.intra_count <= (uint32_t)kIntraMaxTbs && k < nd) ? src[k] : 0u; } }
    };
    if (tid < 35) { s_angle[tid] = c_angle[tid]; s_inv_angle[tid] = c_inv_angle[tid]; }
    // the run's CTB records and the `intra_edge` flags around it (row above: columns c0 - 1 .. c1; this row: c0 - 1), once: the CTB loop looked them up in
    // memory one dependent load at a time -- which CTB is next, its record, whom it waits for
    for (int k = tid; k < (c1 - c0) * 8; k += kIntraThreads) ((uint32_t *)s_ctb)[k] = ((const uint32_t *)(pp.ctbs + cy * pp.ctb_w + c0))[k];
    if (tid < c1 - c0 + 2) { const int col = c0 - 1 + tid;
        s_edge_above[tid] = (cy > 0 && col >= 0 && col < pp.ctb_w) ? pp.ctbs[(cy - 1) * pp.ctb_w + col].intra_edge : (uint8_t)0; }
    if (tid == kIntraThreads - 1) s_edge_above[kIntraMaxRun + 2] = c0 > 0 ? pp.ctbs[cy * pp.ctb_w + c0 - 1].intra_edge : (uint8_t)0;
    __syncthreads();
    auto next_intra = [&](int from) { int c = from; while (c < c1 && !s_ctb[c - c0].intra_count) c++; return c; };
    int cx = next_intra(c0), prev_cx = -2;
    Pre pre;
    if (cx < c1) prefetch(cx, pre);
    while (cx < c1) {
    const HevcCtb ctb = s_ctb[cx - c0];
    const int x0 = cx << pp.ctb_log2, xc0 = x0 >> 1;
    // ---- the left column out of the previous tile, when that CTB was the left neighbour (its samples may still be on their way to memory) ----
    uint32_t keep_ly = pre.ly, keep_lc = pre.lc;
    if (prev_cx == cx - 1) {
        if (tid <= cs) keep_ly = ty[tid * kYS + kYO + cs - 1];
        if (tid <= hc) keep_lc = (uint32_t)tc[0][tid * kCS + kCO + hc - 1] | (uint32_t)tc[1][tid * kCS + kCO + hc - 1] << 8;
    }
    // ---- wait only for the coding tree blocks above (left, straight, right) that themselves hold intra blocks: the others were final before the launch
    //      (one counter per row "everything up to cx + 2 of the row above" chained every intra block of a picture to ALL intra blocks up and to the left) ----
    // ... and of those only the ones whose intra blocks reach their bottom row (HevcCtb.intra_edge): in a P / B picture with scattered intra blocks that
    // leaves few waits at all, and the rows of the picture run side by side instead of as a wavefront.  The CTB to the left matters when it belongs to the
    // run before this one and its intra blocks reach its right column.
    int need[kHevcIntraSegs + 1];
    for (int &v : need) v = 0;
    if (cy > 0) for (int col = cx > 0 ? cx - 1 : 0; col <= cx + 1 &&
        col < pp.ctb_w; col++) if (s_edge_above[col - (c0 - 1)] & 1) need[col / seg_w] = col + 1;
    const bool left_run = cx == c0 && cx > 0 && (s_edge_above[kIntraMaxRun + 2] & 2);
    if (left_run) need[kHevcIntraSegs] = cx;
    if (tid == 0) {
        // Relaxed polls and NO fence: what is handed over -- bottom rows, the last tile of a run -- is written through (st_wt16) before the counter moves,
        // and read with loads that go past the caches, exactly as the rows of an H.264 chain launch hand over their samples (chain_common.h).  An acquire /
        // release pair here made every CTB invalidate and write back the XCD's L2.
        for (int s = 0; s <= kHevcIntraSegs; s++) if (need[s]) {
            const int *ctr = s < kHevcIntraSegs ? &prog[(cy - 1) * kHevcIntraSegs + s] : &prog[cy * kHevcIntraSegs + seg - 1];
            int spins = 0;
            while (ld_coh(ctr) < need[s] && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();                                              // (also: everybody is done with the previous tile)
    const int n_tbs = (int)ctb.intra_count;
    const bool tbs_in_lds = n_tbs <= kIntraMaxTbs;
    if (tbs_in_lds) {
        // (the records came with the CTB's prefetch -- `pre.tb`, 3 dwords per thread --; only a CTB with more than 153 blocks fetches the rest here)
        const uint32_t *src = (const uint32_t *)(pp.itbs + ctb.intra_first); uint32_t *dstw = (uint32_t *)s_tbs;      // 20-byte records, 4-byte aligned
        for (int t = 0; t < 3; t++) { const int k = tid + kIntraThreads * t; if (k < n_tbs * 5) dstw[k] = pre.tb[t]; }
        for (int k = tid + 3 * kIntraThreads; k < n_tbs * 5; k += kIntraThreads) dstw[k] = src[k];
    }
    // ---- the row above (two CTB widths): the one load on the dependency chain ----
    auto ld16_coh = [](const uint8_t *p) {
        const unsigned long long a = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                 b = __hip_atomic_load((const unsigned long long *)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    };
    if (y0 > 0 && tid < 2 * q && x0 + 16 * tid < pp.w) *(uint4 *)&ty[kYO + 16 * tid] = ld16_coh(surf + (size_t)(y0 - 1) * pp.pitch + x0 + 16 * tid);
    if (yc0 > 0 && tid >= 64 && tid < 64 + 2 * q && xc0 + 8 * (tid - 64) < pw) {
        uint2 cb, crv; de_interleave(ld16_coh(cpl + (size_t)(yc0 - 1) * pp.pitch + 2 * (xc0 + 8 * (tid - 64))), cb, crv);
        *(uint2 *)&tc[0][kCO + 8 * (tid - 64)] = cb; *(uint2 *)&tc[1][kCO + 8 * (tid - 64)] = crv;
    }
    if (left_run) {                                               // the left column was written by another workgroup of this launch
        if (tid >= 1 && tid <= cs && y0 + tid - 1 < pp.h) keep_ly = (uint32_t)ld_coh8(surf + (size_t)(y0 + tid - 1) * pp.pitch + x0 - 1);
        if (tid >= 1 && tid <= hc && yc0 + tid - 1 < ph) { const uint8_t *pc = cpl + (size_t)(yc0 + tid - 1) * pp.pitch + 2 * (xc0 - 1);
            keep_lc = (uint32_t)ld_coh8(pc) | (uint32_t)ld_coh8(pc + 1) << 8; }
    }
    // (the corner above-left belongs to a CTB of the row above: it, too, may only be read now)
    if (tid == 128 && x0 > 0 && y0 > 0) keep_ly = (uint32_t)ld_coh8(surf + (size_t)(y0 - 1) * pp.pitch + x0 - 1);
    if (tid == 129 && x0 > 0 && yc0 > 0) { const uint8_t *pc = cpl + (size_t)(yc0 - 1) * pp.pitch + 2 * (xc0 - 1);
        keep_lc = (uint32_t)ld_coh8(pc) | (uint32_t)ld_coh8(pc + 1) << 8; }
    // ---- body and left column from the registers ----
    if (y_mine) *(uint4 *)&ty[(yr + 1) * kYS + kYO + 16 * yg] = pre.y;
    if (c_mine) { uint2 cb, crv; de_interleave(pre.c, cb, crv); *(uint2 *)&tc[0][(cr + 1) * kCS + kCO + 8 * cg] = cb;
        *(uint2 *)&tc[1][(cr + 1) * kCS + kCO + 8 * cg] = crv; }
    for (int t = 0; t < 2; t++) { const int pc = tid + 256 * t; *(uint4 *)&tr_y[(pc >> 3) * 64 + 8 * (pc & 7)] = pre.ry[t]; }
    { int16_t *d = &tr_c[tid >> 7][((tid & 127) >> 2) * 32 + 8 * (tid & 3)]; *(uint2 *)d = pre.rc[0]; *(uint2 *)(d + 4) = pre.rc[1]; }
    if (x0 > 0 && tid >= 1 && tid <= cs) ty[tid * kYS + kYO - 1] = (uint8_t)keep_ly;
    if (x0 > 0 && tid >= 1 && tid <= hc) { tc[0][tid * kCS + kCO - 1] = (uint8_t)keep_lc; tc[1][tid * kCS + kCO - 1] = (uint8_t)(keep_lc >> 8); }
    if (tid == 128 && x0 > 0 && y0 > 0) ty[kYO - 1] = (uint8_t)keep_ly;
    if (tid == 129 && x0 > 0 && yc0 > 0) { tc[0][kCO - 1] = (uint8_t)keep_lc; tc[1][kCO - 1] = (uint8_t)(keep_lc >> 8); }
    // Round 5: the NEXT coding tree block of the run is looked up and fetched here, in front of the block loop (everything fetched was final before the
    // launch, the registers of `pre` are free again): its loads land while this CTB's blocks are predicted.  At the end of the iteration, where round 4 had
    // them, the lookup and the issue cost 3 us per CTB on the picture's dependency chain (probe) and the first use of `pre` waited for memory.
    const int next_cx = next_intra(cx + 1);
    if (next_cx < c1) prefetch(next_cx, pre);
    __syncthreads();
    int lane, nt;
    // ---- block loop: the three colour planes are independent, so wavefront w runs the blocks of plane w on its own (wave-synchronous:
    //      no s_barrier, the phases of a block only cost LDS latency) ----
    const int wave = threadIdx.x >> 6;
    lane = threadIdx.x & 63; nt = 64;
    if (wave < 3) {
    int16_t (*edge)[132] = edge_[wave]; int16_t *refa = refa_[wave];
    const int16_t *rtile = wave ? tr_c[wave - 1] : tr_y; const int rts = wave ? 32 : 64;      // the CTB's residual in LDS
    // this plane's blocks: a wave used to walk the CTB's whole list and skip the other planes' records (one LDS round trip each: ~46 per CTB of an I picture
    // for ~19 luma blocks); it now compacts the indices of its own blocks once (ballot + prefix count)
    int n_mine = 0;
    if (tbs_in_lds) {
        for (int b0 = 0; b0 < n_tbs; b0 += 64) {
            const int idx = b0 + lane;
            const bool mine = idx < n_tbs && s_tbs[idx].plane == wave;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
            if (mine) s_list[wave][n_mine + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)idx;
            n_mine += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // (the NEXT block's record is fetched while this one is predicted: the index list and the record are two dependent LDS round trips)
    const int n_loop = tbs_in_lds ? n_mine : n_tbs;
    auto fetch_tb = [&](int tj) -> HevcIntraTb {
        if (tj >= n_loop) return HevcIntraTb{};
        const int ti = tbs_in_lds ? (int)s_list[wave][tj] : tj;
        return tbs_in_lds ? s_tbs[ti] : pp.itbs[ctb.intra_first + ti];
    };
    HevcIntraTb tv_next = fetch_tb(0);
    for (int tj = 0; tj < n_loop; tj++) {
        // the record is the same for every lane: through v_readfirstlane into scalar registers, so that everything that depends on the block's size and
        // mode only -- loop bounds, the mode dispatch, the tile addresses -- is scalar code and scalar branches (round 5; before, each of those was
        // an exec-mask region of vector compares on the dependency chain of the picture)
        HevcIntraTb tb;
        { uint32_t w[5]; __builtin_memcpy(w, &tv_next, 20);
          for (int i = 0; i < 5; i++) w[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w[i]);
          __builtin_memcpy(&tb, w, 20); }
        tv_next = fetch_tb(tj + 1);
        if (tb.plane != wave) continue;
        const int log2 = tb.log2, n = 1 << log2, c = wave, N = 4 * n, unit = c ? 2 : 4, ush = c ? 1 : 2;
        const bool pcm = tb.mode == kHevcModePcm;
        uint8_t *tile = c ? tc[c - 1] : ty; const int ts = c ? kCS : kYS;
        const int lx = tb.x - (c ? x0 >> 1 : x0) + (c ? kCO : kYO), ly = tb.y - (c ? y0 >> 1 : y0) + 1;     // block origin inside the tile
        const int16_t *rblk = rtile + (tb.y - (c ? y0 >> 1 : y0)) * rts + tb.x - (c ? x0 >> 1 : x0);      // the block's residual (k_hevc_iresid)
        const int16_t *e = edge[0];
        int dc_fused = 0; bool have_dc = false;
        if (!pcm) {
            // ---- neighbouring samples with substitution (8.4.4.2.2), every entry on its own lane: availability comes in units of 4 luma
            //      samples, so the substitute of an unavailable entry is found with bit scans over a mask of <= 33 segments ----
            const int U = (2 * n) >> ush;
            uint64_t mask = __brev(tb.avail << (32 - U));                // left units bottom to top: bit j = avail bit U - 1 - j (U <= 16)
            if (tb.flags & HTB_CORNER) mask |= 1ull << U;
            mask |= (uint64_t)((tb.avail >> 16) & ((1u << U) - 1)) << (U + 1);
            const bool full = mask == ((2ull << (2 * U)) - 1);
            if (full) {
                // Every neighbour is available (the usual case inside an I picture): no substitution, so the smoothing filter (8.4.4.2.3) and the DC sum need
                // not wait for a gathered copy of the edge -- a lane reads entry i and its two neighbours straight from the tile.  One phase (and one LDS
                // round trip on the block's dependency chain) instead of gather -> barrier -> filter -> barrier -> DC sum (round 5).
                auto raw = [&](int j) -> int { const int d = j - 2 * n, trow = d < 0 ? ly - 1 - d : ly - 1, tcol = d <= 0 ? lx - 1 : lx - 1 + d;
                    return tile[trow * ts + tcol]; };
                bool filt = false, strong = false;
                if (c == 0 && tb.mode != 1 && n > 4) {
                    const int dv = iabs(tb.mode - 26), dh = iabs(tb.mode - 10), md = dv < dh ? dv : dh, thr = n == 8 ? 7 : (n == 16 ? 1 : 0);
                    filt = md > thr;
                }
                int e0 = 0, e64 = 0, e128 = 0;
                if (filt && n == 32 && pp.strong_intra) { e0 = raw(0); e64 = raw(64); e128 = raw(128);
                    strong = iabs(e64 + e128 - 2 * raw(96)) < 8 && iabs(e64 + e0 - 2 * raw(32)) < 8; }
                int dcp = 0;
                for (int i = lane; i <= N; i += nt) {
                    const int v1 = raw(i);
                    int v = v1;
                    if (filt && i != 0 && i != N) {
                        if (strong) v = i == 64 ? e64 : (i < 64 ? (i * e64 + (64 - i) * e0 + 32) >> 6 : ((128 - i) * e64 + (i - 64) * e128 + 32) >> 6);
                        else v = (raw(i - 1) + 2 * v1 + raw(i + 1) + 2) >> 2;
                    }
                    if ((i >= n && i < 2 * n) || (i > 2 * n && i <= 3 * n)) dcp += v1;      // L[0 .. n-1] and T[0 .. n-1] (used by DC, which is never smoothed)
                    edge[0][i] = (int16_t)v;
                }
                if (tb.mode == 1) { dc_fused = (n + wave_sum(dcp)) >> (log2 + 1); have_dc = true; }
                __builtin_amdgcn_wave_barrier();
            } else {
            for (int i = lane; i <= N; i += nt) {
                int v = 128;
                if (mask) {
                    const int sgm = i < 2 * n ? i >> ush : (i == 2 * n ? U : U + 1 + ((i - 2 * n - 1) >> ush));
                    int src = i;
                    if (!full && !((mask >> sgm) & 1)) {                 // (full: every neighbour is available -- the usual case inside an I picture)
                        const uint64_t below = mask & ((1ull << sgm) - 1);
                        if (below) { const int t = 63 - __clzll((long long)below);
                            src = t < U ? t * unit + unit - 1 : (t == U ? 2 * n : 2 * n + 1 + (t - U - 1) * unit + unit - 1); }
                        else { const int f = __ffsll((long long)mask) - 1; src = f < U ? f * unit : (f == U ? 2 * n : 2 * n + 1 + (f - U - 1) * unit); }
                    }
                    // entry src: left column bottom to top (src < 2n), corner (2n), top row left to right -- one read at (row, column), no branches
                    const int d = src - 2 * n, trow = d < 0 ? ly - 1 - d : ly - 1, tcol = d <= 0 ? lx - 1 : lx - 1 + d;
                    v = tile[trow * ts + tcol];
                }
                edge[0][i] = (int16_t)v;
            }
            __builtin_amdgcn_wave_barrier();
            // ---- filtering (8.4.4.2.3) ----
            if (c == 0 && tb.mode != 1 && n > 4) {
                const int dv = iabs(tb.mode - 26), dh = iabs(tb.mode - 10), md = dv < dh ? dv : dh, thr = n == 8 ? 7 : (n == 16 ? 1 : 0);
                if (md > thr) {
                    const bool strong = pp.strong_intra && n == 32 && iabs(edge[0][64] + edge[0][128] - 2 * edge[0][96]) < 8 &&
                        iabs(edge[0][64] + edge[0][0] - 2 * edge[0][32]) < 8;
                    for (int i = lane; i <= N; i += nt) {
                        int v;
                        if (i == 0 || i == N) v = edge[0][i];
                        else if (strong) v = i == 64 ? edge[0][64] :
                            (i < 64 ? (i * edge[0][64] + (64 - i) * edge[0][0] + 32) >> 6 : ((128 - i) * edge[0][64] + (i - 64) * edge[0][128] + 32) >> 6);
                        else v = (edge[0][i - 1] + 2 * edge[0][i] + edge[0][i + 1] + 2) >> 2;
                        edge[1][i] = (int16_t)v;
                    }
                    e = edge[1];
                }
            }
            __builtin_amdgcn_wave_barrier();
            }   // !full
        }
        const int16_t *L = e + 2 * n - 1, *T = e + 2 * n + 1;      // L[-y] = left sample of row y, T[x] = top sample of column x, T[-1] = corner
        int ang = 0; bool vert = false;
        int dc = 0;
        if (!pcm && tb.mode == 1) dc = have_dc ? dc_fused : (n + wave_sum(lane < n ? L[-lane] + T[lane] : 0)) >> (log2 + 1);   // (n <= 32: one entry of each edge per lane)
        if (!pcm && tb.mode >= 2) {
            ang = s_angle[tb.mode]; vert = tb.mode >= 18;
            // the main reference array ref[] of 8.4.4.2.6 is the edge itself when the angle is not negative (ref[i] = T[i - 1] resp. L[-(i - 1)], i >= 0):
            // only negative angles, which extend it below index 0 with projected samples of the other edge, build it in LDS (round 5)
            if (ang < 0) {
                const int inv = s_inv_angle[tb.mode], lo = (n * ang) >> 5;
                for (int i = lo + lane; i <= 2 * n; i += nt) {
                    int v;
                    if (i >= 0) v = vert ? T[i - 1] : L[-(i - 1)];
                    else { const int k = (i * inv + 128) >> 8; v = vert ? L[-(k - 1)] : T[k - 1]; }
                    refa[32 + i] = (int16_t)v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ref[i] == rbase[rs * i]
        const int16_t *rbase = ang < 0 ? refa + 32 : (vert ? T - 1 : L + 1);
        const int rs = (ang < 0 || vert) ? 1 : -1;
        // Round 5: a lane predicts FOUR neighbouring samples of a row and stores them as one dword (block positions are multiples of 4 in the tile).  A
        // 32x32 block is 4 rounds of the wave instead of 16, each round's LDS reads (references, residual) are in flight together -- the byte-wise stores of
        // the one-sample form may alias anything, so the compiler kept every round's reads behind the previous round's store: one LDS round trip per round
        // on the dependency chain (probe: the prediction loop was half of a block's 2.9 us).
        const int qsh = log2 - 2, nq = n << qsh, iters = (nq + 63) >> 6;      // quads per row = n / 4; n == 4: four lanes of one round
#pragma unroll 1
        for (int i = 0; i < iters; i++) {
            const int k = lane + 64 * i;
            if (k >= nq) break;
            const int y = k >> qsh, x = (k & ((1 << qsh) - 1)) << 2;
            int v[4];
            if (pcm) { v[0] = v[1] = v[2] = v[3] = 0; }
            else if (tb.mode == 0) {
                const int ly_ = L[-y], tn = T[n], ln = L[-n];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = ((n - 1 - (x + j)) * ly_ + (x + j + 1) * tn + (n - 1 - y) * T[x + j] + (y + 1) * ln + n) >> (log2 + 1);
            } else if (tb.mode == 1) {
                v[0] = v[1] = v[2] = v[3] = dc;
                if (c == 0 && n < 32) {
                    if (y == 0) {
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] = (T[x + j] + 3 * dc + 2) >> 2;
                        if (x == 0) v[0] = (L[0] + 2 * dc + T[0] + 2) >> 2;
                    } else if (x == 0) v[0] = (L[-y] + 3 * dc + 2) >> 2;
                }
            } else if (vert) {
                // the row's offset and fraction are shared by its samples: five neighbouring references for four samples
                const int pos = (y + 1) * ang, idx = pos >> 5, fr = pos & 31;
                const int16_t *rp = rbase + x + idx + 1;                     // (vertical modes: rs == 1)
                const int r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3], r4 = rp[4];
                if (fr) { v[0] = ((32 - fr) * r0 + fr * r1 + 16) >> 5; v[1] = ((32 - fr) * r1 + fr * r2 + 16) >> 5; v[2] = ((32 - fr) * r2 + fr * r3 + 16) >> 5;
                    v[3] = ((32 - fr) * r3 + fr * r4 + 16) >> 5; }
                else { v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3; }
                if (c == 0 && n < 32 && ang == 0 && x == 0) v[0] = clip1(T[0] + ((L[-y] - T[-1]) >> 1));
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int pos = (x + j + 1) * ang, idx = pos >> 5, fr = pos & 31;
                    const int r0 = rbase[rs * (y + idx + 1)], r1 = rbase[rs * (y + idx + 2)];
                    v[j] = fr ? ((32 - fr) * r0 + fr * r1 + 16) >> 5 : r0;
                }
                if (c == 0 && n < 32 && ang == 0 && y == 0) {
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = clip1(L[0] + ((T[x + j] - T[-1]) >> 1));
                }
            }
            if (tb.coef_n) { const uint2 rr = *(const uint2 *)(rblk + y * rts + x);
                v[0] = clip1(v[0] + (int)(short)(rr.x & 0xffffu)); v[1] = clip1(v[1] + ((int)rr.x >> 16)); v[2] = clip1(v[2] + (int)(short)(rr.y & 0xffffu));
                v[3] = clip1(v[3] + ((int)rr.y >> 16)); }
            *(uint32_t *)&tile[(ly + y) * ts + lx + x] = (uint32_t)v[0] | (uint32_t)v[1] << 8 | (uint32_t)v[2] << 16 | (uint32_t)v[3] << 24;
        }
        __builtin_amdgcn_wave_barrier();
    }
    }   // wave < 3
    __syncthreads();
    // ---- the last CTB of a run whose intra blocks reach its right column: the next run reads that column -- the whole tile is written through ----
    const bool wt_all = (ctb.intra_edge & 2) && cx == c1 - 1 && c1 < pp.ctb_w;
    if (wt_all) {
        if (y_mine && y0 + yr < pp.h && x0 + 16 * yg < pp.w) st_wt16(surf + (size_t)(y0 + yr) * pp.pitch + x0 + 16 * yg,
            *(const uint4 *)&ty[(yr + 1) * kYS + kYO + 16 * yg]);
        if (c_mine && yc0 + cr < ph && xc0 + 8 * cg < pw)
            st_wt16(cpl + (size_t)(yc0 + cr) * pp.pitch + 2 * (xc0 + 8 * cg), interleave(*(const uint2 *)&tc[0][(cr + 1) * kCS + kCO + 8 * cg],
                *(const uint2 *)&tc[1][(cr + 1) * kCS + kCO + 8 * cg]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) st_coh(&prog[cy * kHevcIntraSegs + seg], cx + 1);
    } else {
    // ---- the bottom row of the CTB first: it is all the row below needs of this CTB; then the counter ----
    if (tid < q) { const int y = y0 + cs - 1; if (y < pp.h && x0 + 16 * tid < pp.w) st_wt16(surf + (size_t)y * pp.pitch + x0 + 16 * tid,
        *(const uint4 *)&ty[cs * kYS + kYO + 16 * tid]); }
    else if (tid < 2 * q) { const int g = tid - q, y = yc0 + hc - 1, x = xc0 + 8 * g;
        if (y < ph && x < pw) st_wt16(cpl + (size_t)y * pp.pitch + 2 * x, interleave(*(const uint2 *)&tc[0][hc * kCS + kCO + 8 * g],
            *(const uint2 *)&tc[1][hc * kCS + kCO + 8 * g])); }
    // (wave 0 holds every one of those write-through stores: once they are complete the counter may move)
    if (tid < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) st_coh(&prog[cy * kHevcIntraSegs + seg], cx + 1);
    // ---- the rest of the tile (its intra blocks changed; the other samples are written back unchanged), off the chain ----
    if (y_mine && yr < cs - 1 && y0 + yr < pp.h &&
        x0 + 16 * yg < pp.w) *(uint4 *)(surf + (size_t)(y0 + yr) * pp.pitch + x0 + 16 * yg) = *(const uint4 *)&ty[(yr + 1) * kYS + kYO + 16 * yg];
    if (c_mine && cr < hc - 1 && yc0 + cr < ph && xc0 + 8 * cg < pw)
        *(uint4 *)(cpl + (size_t)(yc0 + cr) * pp.pitch + 2 * (xc0 + 8 * cg)) = interleave(*(const uint2 *)&tc[0][(cr + 1) * kCS + kCO + 8 * cg],
            *(const uint2 *)&tc[1][(cr + 1) * kCS + kCO + 8 * cg]);
    }
    // rows of a CTB that reaches below the picture: the "bottom row" above lay outside, the last rows inside were written just now -- nobody waits for them
    prev_cx = cx;
    cx = next_cx;
    }   // CTBs of the row
}

// ------------------------------------------------------------------------------------------------------------
// 8.7.2.3 / 8.7.2.4: which edges are filtered and how strongly -- on the device since round 5 (the host spent 9 % of its parse time per picture on it)
// ------------------------------------------------------------------------------------------------------------
// The job lists already say everything the derivation needs: where the prediction blocks lie and what their motion is (HevcPu), where the intra blocks lie
// (HevcIntraTb, luma entries) and where the luma transform blocks with cbf_luma = 1 lie (HevcTb, luma entries).  k_hevc_bs_raster paints them into maps of
// 4x4 cells; k_hevc_bs reads the maps per 4-sample edge segment of the 8x8 grid and writes bs_v / bs_h in the layout k_hevc_deblock has always read:
//   an edge exists where a painted block ends (flag planes 2 / 3: the left / top edge of a cell is the edge of an intra or coded block -- painted on the
//   block's own first column / row AND on the cells behind its last ones) or where two different prediction blocks meet;
//   bS 2: p0 or q0 intra; 1: a transform edge with coefficients on either side, or different motion (8.7.2.4); else 0.
// (The lists hold prediction blocks cut into pieces of at most 16x16 and no transform blocks without coefficients: edges between pieces of one
// prediction block compare equal motion and give 0, and a transform edge between two blocks without coefficients inside one prediction block gives 0 by
// the clause itself.)  What the lists cannot say is in HevcCtb.db_flags.  All plain stores of the value 1 / of a block's own index: no atomics needed.
__global__ __launch_bounds__(256) void k_hevc_bs_clear(const HevcPicParams *pics) {
    typedef const __attribute__((address_space(4))) HevcPicParams ConstPic;
    ConstPic &pp = *(ConstPic *)(uintptr_t)(pics + blockIdx.y);
    if (!(pp.stages & HPS_DEBLOCK)) return;
    const int n16 = ((pp.w >> 2) * (pp.h >> 2)) >> 2;                // 4 planes x cells bytes = cells / 4 uint4
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) ((uint4 *)pp.cell_flags)[i] = make_uint4(0, 0, 0, 0);
}
__global__ __launch_bounds__(256) void k_hevc_bs_raster(const HevcPicParams *pics) {
    typedef const __attribute__((address_space(4))) HevcPicParams ConstPic;
    ConstPic &pp = *(ConstPic *)(uintptr_t)(pics + blockIdx.y);
    if (!(pp.stages & HPS_DEBLOCK)) return;
    const hbs::Maps m = hbs::maps_of(pp.w, pp.h, pp.pu_map, pp.cell_flags);
    const int gid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int i = gid; i < pp.n_pus; i += stride) hbs::paint_pu(m, i, pp.pus[i]);
    for (int i = gid; i < pp.n_itbs; i += stride) { const HevcIntraTb tb = pp.itbs[i]; if (tb.plane == 0) hbs::mark(m, tb.x, tb.y, 1 << tb.log2, m.f_intra); }
    for (int i = gid; i < pp.n_tbs; i += stride) { const HevcTb tb = pp.tbs[i]; if (tb.plane == 0) hbs::mark(m, tb.x, tb.y, 1 << tb.log2, m.f_cbf); }
}
// blockIdx.z = direction: 0 vertical edges (x = 8k, one entry per 4 rows), 1 horizontal edges (y = 8k, one entry per 4 columns)
__global__ __launch_bounds__(256) void k_hevc_bs(const HevcPicParams *pics) {
    typedef const __attribute__((address_space(4))) HevcPicParams ConstPic;
    ConstPic &pp = *(ConstPic *)(uintptr_t)(pics + blockIdx.y);
    if (!(pp.stages & HPS_DEBLOCK)) return;
    const int dir = (int)blockIdx.z;
    const int w8 = pp.w >> 3, w4 = pp.w >> 2, h4 = pp.h >> 2, h8 = pp.h >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    int x, y;
    if (dir == 0) { if (idx >= w8 * h4) return; x = (idx % w8) * 8; y = (idx / w8) * 4; }
    else { if (idx >= w4 * h8) return; x = (idx % w4) * 4; y = (idx / w4) * 8; }
    const hbs::Maps m = hbs::maps_of(pp.w, pp.h, pp.pu_map, pp.cell_flags);
    (dir ? pp.bs_h : pp.bs_v)[idx] = (uint8_t)hbs::edge_strength(m, dir, x, y, pp.ctb_log2, pp.ctb_w, pp.ctbs, pp.pus, pp.qp8, pp.w8);
}

// ------------------------------------------------------------------------------------------------------------
// 8.7.2.5: deblocking, one lane per 4-sample edge segment.  dir 0: vertical edges (filter across x), dir 1: horizontal edges
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hevc_deblock(const HevcPicParams *pics, int dir) {
    const HevcPicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & HPS_DEBLOCK)) return;
    const int w8 = pp.w >> 3, w4 = pp.w >> 2, h4 = pp.h >> 2, h8 = pp.h >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    int x, y, b;
    if (dir == 0) { if (idx >= w8 * h4) return; x = (idx % w8) * 8; y = (idx / w8) * 4; b = pp.bs_v[idx]; }
    else { if (idx >= w4 * h8) return; x = (idx % w4) * 4; y = (idx / w4) * 8; b = pp.bs_h[idx]; }
    const int bs = b & 3;
    if (!bs) return;
    const bool keep_p = b & 4, keep_q = b & 8;
    const int xp = dir ? x : x - 1, yp = dir ? y - 1 : y;
    const int qpq = pp.qp8[(y >> 3) * pp.w8 + (x >> 3)] & 63, qpp = pp.qp8[(yp >> 3) * pp.w8 + (xp >> 3)] & 63, qpl = (qpq + qpp + 1) >> 1;
    const HevcCtb &cq = pp.ctbs[(y >> pp.ctb_log2) * pp.ctb_w + (x >> pp.ctb_log2)];
    uint8_t *surf = pp.work_surf;
    {
        const int beta = c_beta[clip3(0, 51, qpl + 2 * cq.beta_off)], tc = c_tc[clip3(0, 53, qpl + 2 * (bs - 1) + 2 * cq.tc_off)];
        const int across = dir ? pp.pitch : 1, along = dir ? 1 : pp.pitch;
        uint8_t *q = surf + (size_t)y * pp.pitch + x;
        int s[4][8];                                                    // s[line][0..7] = p3 p2 p1 p0 q0 q1 q2 q3
        for (int k = 0; k < 4; k++) for (int i = 0; i < 8; i++) s[k][i] = q[k * along + (i - 4) * across];
        const int dp0 = iabs(s[0][1] - 2 * s[0][2] + s[0][3]), dp3 = iabs(s[3][1] - 2 * s[3][2] + s[3][3]), dq0 = iabs(s[0][6] - 2 * s[0][5] + s[0][4]),
            dq3 = iabs(s[3][6] - 2 * s[3][5] + s[3][4]);
        if (dp0 + dq0 + dp3 + dq3 < beta) {
            bool strong = true;
            for (int k = 0; k < 4; k += 3) { const int dk = k ? dp3 + dq3 : dp0 + dq0;
                if (!(2 * dk < (beta >> 2) && iabs(s[k][0] - s[k][3]) + iabs(s[k][4] - s[k][7]) < (beta >> 3) &&
                iabs(s[k][3] - s[k][4]) < ((5 * tc + 1) >> 1))) strong = false; }
            const int side = (beta + (beta >> 1)) >> 3; const bool mp = dp0 + dp3 < side, mq = dq0 + dq3 < side;
            for (int k = 0; k < 4; k++) {
                const int p3 = s[k][0], p2 = s[k][1], p1 = s[k][2], p0 = s[k][3], q0 = s[k][4], q1 = s[k][5], q2 = s[k][6], q3 = s[k][7];
                uint8_t *l = q + k * along;
                if (strong) {
                    if (!keep_p) { l[-across] = (uint8_t)clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
                        l[-2 * across] = (uint8_t)clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
                        l[-3 * across] = (uint8_t)clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3); }
                    if (!keep_q) { l[0] = (uint8_t)clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
                        l[across] = (uint8_t)clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
                        l[2 * across] = (uint8_t)clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3); }
                } else {
                    int dl = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
                    if (iabs(dl) >= 10 * tc) continue;
                    dl = clip3(-tc, tc, dl);
                    if (!keep_p) { l[-across] = (uint8_t)clip1(p0 + dl); if (mp) l[-2 * across] = (uint8_t)clip1(p1 + clip3(-(tc >> 1), tc >> 1,
                        (((p2 + p0 + 1) >> 1) - p1 + dl) >> 1)); }
                    if (!keep_q) { l[0] = (uint8_t)clip1(q0 - dl); if (mq) l[across] = (uint8_t)clip1(q1 + clip3(-(tc >> 1), tc >> 1,
                        (((q2 + q0 + 1) >> 1) - q1 - dl) >> 1)); }
                }
            }
        }
    }
    // chroma (8.7.2.5.5): edges on the 8-sample chroma grid, bS 2 only, four chroma lines per unit taken at the unit's first luma segment
    if (bs == 2 && ((dir ? y : x) & 15) == 0 && ((dir ? x : y) & 7) == 0) {
        const int xc = x >> 1, yc = y >> 1;
        for (int c = 1; c < 3; c++) {
            const int qpc = c_qpc[clip3(0, 57, qpl + (c == 1 ? pp.cb_qp_off : pp.cr_qp_off))], tc = c_tc[clip3(0, 53, qpc + 2 + 2 * cq.tc_off)];
            for (int k = 0; k < 4; k++) {
                uint8_t *p1 = sample_ptr(surf, pp, c, dir ? xc + k : xc - 2, dir ? yc - 2 : yc + k),
                    *p0 = sample_ptr(surf, pp, c, dir ? xc + k : xc - 1, dir ? yc - 1 : yc + k);
                uint8_t *q0 = sample_ptr(surf, pp, c, dir ? xc + k : xc, dir ? yc : yc + k),
                    *q1 = sample_ptr(surf, pp, c, dir ? xc + k : xc + 1, dir ? yc + 1 : yc + k);
                const int dl = clip3(-tc, tc, (((*q0 - *p0) << 2) + *p1 - *q1 + 4) >> 3), np = clip1(*p0 + dl), nq = clip1(*q0 - dl);
                if (!keep_p) *p0 = (uint8_t)np;
                if (!keep_q) *q0 = (uint8_t)nq;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// 8.7.3: sample adaptive offset, work surface -> current surface
// ------------------------------------------------------------------------------------------------------------
// Round 5: one lane per DWORD of a surface row -- four luma samples, or two Cb Cr pairs of the interleaved chroma plane (both chroma components share the
// SAO type and the edge class, 7.3.8.3; offsets and band positions are their own).  Rounds 1-4 ran one lane per SAMPLE with byte loads, byte stores and the
// CTB record fetched per sample: 22 M wave-instructions and 71 us per 4K picture, ~0.2 TB/s for a kernel that only streams a surface.  Here the neighbours of
// the edge classes come from the row's own dwords and the rows above / below with v_alignbyte (sample distance 1 byte in luma, 2 in chroma).
// grid: x = 256-byte pieces of a row, y = picture, z = four rows (luma rows first, then the chroma rows); block = 4 rows x 64 lanes.
__global__ __launch_bounds__(256) void k_hevc_sao(const HevcPicParams *pics) {
    typedef const __attribute__((address_space(4))) HevcPicParams ConstPic;
    ConstPic &pp = *(ConstPic *)(uintptr_t)(pics + blockIdx.y);
    if (!(pp.stages & HPS_SAO)) return;
    const int xw = (int)blockIdx.x * 64 + (int)(threadIdx.x & 63), row = (int)blockIdx.z * 4 + (int)(threadIdx.x >> 6);
    const int w = pp.w, h = pp.h;
    if (4 * xw >= w || row >= h + (h >> 1)) return;
    const bool chroma = row >= h;
    const int y = chroma ? row - h : row, ph = chroma ? h >> 1 : h;                    // row inside its plane, rows of the plane
    const int xl = 4 * xw, yl = chroma ? 2 * y : y;                                   // luma position of the dword's first sample
    const int lg = pp.ctb_log2, cs = 1 << lg, cxb = xl >> lg, cyb = yl >> lg;
    const size_t off = (size_t)(chroma ? pp.chroma_offset : 0) + (size_t)y * pp.pitch + xl;
    const uint8_t *src = pp.work_surf + off;
    uint32_t *dstw = (uint32_t *)(pp.surf[pp.cur] + off);
    const uint32_t cur = *(const uint32_t *)src;
    const uint32_t *cw = (const uint32_t *)(pp.ctbs + cyb * pp.ctb_w + cxb);         // HevcCtb: type[3] pos[3] off[3][4] beta tc nb_mask ...
    const uint32_t w0 = cw[0], w1 = cw[1];
    const int c0 = chroma ? 1 : 0;
    const int type = (int)((w0 >> (8 * c0)) & 255u);
    if (!type || (pp.qp8[(yl >> 3) * pp.w8 + (xl >> 3)] & 128)) { *dstw = cur; return; }      // SAO off here, or samples exempt from the loop filters
    const uint32_t w2 = cw[2], w3 = cw[3], w4 = cw[4];
    // offsets of this lane's component(s), four signed bytes each: component c sits at bytes 6 + 4c .. of the record
    const uint32_t offA = chroma ? __builtin_amdgcn_alignbyte(w3, w2, 2) : __builtin_amdgcn_alignbyte(w2, w1, 2);     // luma / Cb
    const uint32_t offB = __builtin_amdgcn_alignbyte(w4, w3, 2);                                                       // Cr
    const int posA = chroma ? (int)(w1 & 255u) : (int)(w0 >> 24), posB = (int)((w1 >> 8) & 255u);                      // band position / edge class
    int add[4] = {0, 0, 0, 0};
    if (type == 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool second = chroma && (j & 1);
            const int v = (int)((cur >> (8 * j)) & 255u), k = ((v >> 3) - (second ? posB : posA)) & 31;
            if (k < 4) add[j] = (int)(int8_t)((second ? offB : offA) >> (8 * k));
        }
    } else {
        // edge offset: a = the neighbour towards (-dx, -dy), b = its mirror image; s = bytes between neighbouring samples of one component
        const int cls = posA, s = chroma ? 2 : 1;
        const int dx = cls == 1 ? 0 : (cls == 3 ? -1 : 1), dy = cls == 0 ? 0 : 1;
        const int nb = (int)(cw[5] & 255u);
        const bool row_a_ok = y - dy >= 0, row_b_ok = y + dy < ph;
        const uint8_t *ra = src - (dy && row_a_ok ? pp.pitch : 0), *rb = src + (dy && row_b_ok ? pp.pitch : 0);
        // the dwords around the lane's own in the rows of a and b (clamped at the row's ends: such samples are not modified, below)
        const bool has_l = xw > 0, has_r = 4 * xw + 4 < w;
        uint32_t A, B;
        if (dx == 0) { A = *(const uint32_t *)ra; B = *(const uint32_t *)rb; }
        else {
            const uint32_t a_c = *(const uint32_t *)ra, b_c = *(const uint32_t *)rb;
            const uint32_t a_l = has_l ? *(const uint32_t *)(ra - 4) : 0u, a_r = has_r ? *(const uint32_t *)(ra + 4) : 0u;
            const uint32_t b_l = has_l ? *(const uint32_t *)(rb - 4) : 0u, b_r = has_r ? *(const uint32_t *)(rb + 4) : 0u;
            // a lies at x - dx: to the left (bytes x - s ..) for dx = 1, to the right for dx = -1; b the other way round
            const uint32_t a_left = __builtin_amdgcn_alignbyte(a_c, a_l, 4 - s), a_right = __builtin_amdgcn_alignbyte(a_r, a_c, s);
            const uint32_t b_left = __builtin_amdgcn_alignbyte(b_c, b_l, 4 - s), b_right = __builtin_amdgcn_alignbyte(b_r, b_c, s);
            A = dx > 0 ? a_left : a_right; B = dx > 0 ? b_right : b_left;
        }
        const int pw = chroma ? w >> 1 : w, sc = chroma ? 1 : 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool second = chroma && (j & 1);
            const int xs = chroma ? 2 * xw + (j >> 1) : 4 * xw + j;                     // sample column in its plane
            const int xa = xs - dx, ya = y - dy, xb = xs + dx, yb = y + dy;
            bool ok = xa >= 0 && xb >= 0 && xa < pw && xb < pw && ya >= 0 && yb < ph;
            if (ok) {
                // neighbours in another CTB: allowed only where the host's slice / tile analysis says so (HevcCtb.nb_mask: L R T B TL TR BL BR)
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int xn = (k ? xb : xa) << sc, yn = (k ? yb : ya) << sc, ddx = (xn >> lg) - cxb, ddy = (yn >> lg) - cyb;
                    if (ddx == 0 && ddy == 0) continue;
                    const int dirk = ddy == 0 ? (ddx < 0 ? 0 : 1) : (ddx == 0 ? (ddy < 0 ? 2 : 3) : (ddy < 0 ? (ddx < 0 ? 4 : 5) : (ddx < 0 ? 6 : 7)));
                    if (!((nb >> dirk) & 1)) ok = false;
                }
            }
            if (ok) {
                const int v = (int)((cur >> (8 * j)) & 255u), a = (int)((A >> (8 * j)) & 255u), b = (int)((B >> (8 * j)) & 255u);
                const int sg = (v > a) - (v < a) + (v > b) - (v < b);
                const uint32_t ow = second ? offB : offA;
                add[j] = sg == -2 ? (int)(int8_t)ow : sg == -1 ? (int)(int8_t)(ow >> 8) : sg == 1 ? (int)(int8_t)(ow >> 16) : sg == 2 ? (int)(int8_t)(ow >> 24) : 0;
            }
        }
        (void)cs;
    }
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) out |= (uint32_t)clip1((int)((cur >> (8 * j)) & 255u) + add[j]) << (8 * j);
    *dstw = out;
}

// ------------------------------------------------------------------------------------------------------------
static void upload_tables() {
    static bool done[64] = {false};
    int dev = 0; hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || done[dev]) return;
    hipMemcpyToSymbol(HIP_SYMBOL(c_lf), hevc_luma_filter, sizeof c_lf); hipMemcpyToSymbol(HIP_SYMBOL(c_cf), hevc_chroma_filter, sizeof c_cf);
    hipMemcpyToSymbol(HIP_SYMBOL(c_angle), hevc_intra_angle, sizeof c_angle); hipMemcpyToSymbol(HIP_SYMBOL(c_inv_angle), hevc_inv_angle, sizeof c_inv_angle);
    hipMemcpyToSymbol(HIP_SYMBOL(c_beta), hevc_beta_tab, sizeof c_beta); hipMemcpyToSymbol(HIP_SYMBOL(c_tc), hevc_tc_tab, sizeof c_tc);
    hipMemcpyToSymbol(HIP_SYMBOL(c_qpc), hevc_qpc_tab, sizeof c_qpc);
    {   // the transform matrices as 16-bit pairs (hevc_resid_packed.h)
        static uint32_t mp[hrp::kPairDw];
        hrp::build_pair_table(hevc_trans, hevc_dst, mp);
        hipMemcpyToSymbol(HIP_SYMBOL(g_resid_pairs), mp, sizeof mp);
    }
    {   // the interpolation taps in the forms hevc_mc_packed.h works with
        uint32_t lh[4][2], lv[4][2][5], ch[8], cv[8][2][3];
        for (int f = 0; f < 4; f++) { hpk::luma_taps_h(f, lh[f][0], lh[f][1]); hpk::luma_taps_v(f, false, lv[f][0]); hpk::luma_taps_v(f, true, lv[f][1]); }
        for (int f = 0; f < 8; f++) { ch[f] = hpk::chroma_taps_h(f); hpk::chroma_taps_v(f, false, cv[f][0]); hpk::chroma_taps_v(f, true, cv[f][1]); }
        hipMemcpyToSymbol(HIP_SYMBOL(c_lh), lh, sizeof lh); hipMemcpyToSymbol(HIP_SYMBOL(c_lv), lv, sizeof lv);
        hipMemcpyToSymbol(HIP_SYMBOL(c_ch), ch, sizeof ch); hipMemcpyToSymbol(HIP_SYMBOL(c_cv), cv, sizeof cv);
    }
    done[dev] = true;
}
void hevc_kernels_init() { upload_tables(); }

void launch_hevc_picture_batch(const HevcPicParams *d_pics, int n, const HevcBatchDims &m, int *progress, hipStream_t st, hipEvent_t *marks) {
    upload_tables();
    if (marks) hipEventRecord(marks[0], st);
    if (m.max_pus > 0) hipLaunchKernelGGL(k_hevc_mc, dim3(((m.max_pus + kMcWaves - 1) / kMcWaves + 7) & ~7, n), dim3(64 * kMcWaves), 0, st, d_pics);
    constexpr int per_wg = kResidWaves * kResidPerWave;
    if (m.max_tbs > 0) hipLaunchKernelGGL(k_hevc_resid, dim3(((m.max_tbs + per_wg - 1) / per_wg + 7) & ~7, n), dim3(64 * kResidWaves), 0, st, d_pics);
    constexpr int iper_wg = kResidWaves * kIresidPerWave;
    if (m.any_intra && m.max_itbs > 0) hipLaunchKernelGGL(k_hevc_iresid, dim3((m.max_itbs + iper_wg - 1) / iper_wg, n), dim3(64 * kResidWaves), 0, st, d_pics);
    if (marks) hipEventRecord(marks[1], st);
    if (m.any_intra) {
        hipMemsetAsync(progress, 0, sizeof(int) * (size_t)n * kHevcProgressStride, st);
        hipLaunchKernelGGL(k_hevc_intra, dim3(m.max_ctb_h * kHevcIntraSegs, n), dim3(kIntraThreads), 0, st, d_pics, progress, kHevcProgressStride);
    }
    if (marks) hipEventRecord(marks[2], st);
    if (m.any_deblock) {
        // boundary strengths (k_hevc_bs_*): the maps are painted from the job lists, the strengths read the maps -- three small launches in front of the filter
        const int cells = (m.max_w >> 2) * (m.max_h >> 2), entries = m.max_pus > m.max_tbs ? (m.max_pus > m.max_itbs ? m.max_pus : m.max_itbs)
            : (m.max_tbs > m.max_itbs ? m.max_tbs : m.max_itbs);
        hipLaunchKernelGGL(k_hevc_bs_clear, dim3((cells / 4 + 255) / 256, n), dim3(256), 0, st, d_pics);
        hipLaunchKernelGGL(k_hevc_bs_raster, dim3((entries + 255) / 256 > 0 ? (entries + 255) / 256 : 1, n), dim3(256), 0, st, d_pics);
        hipLaunchKernelGGL(k_hevc_bs, dim3((cells / 2 + 255) / 256, n, 2), dim3(256), 0, st, d_pics);
        hipLaunchKernelGGL(k_hevc_deblock, dim3(((m.max_w >> 3) * (m.max_h >> 2) + 255) / 256, n), dim3(256), 0, st, d_pics, 0);
        hipLaunchKernelGGL(k_hevc_deblock, dim3(((m.max_w >> 2) * (m.max_h >> 3) + 255) / 256, n), dim3(256), 0, st, d_pics, 1);
    }
    if (m.any_sao) hipLaunchKernelGGL(k_hevc_sao, dim3((m.max_w / 4 + 63) / 64, n, (m.max_h + m.max_h / 2 + 3) / 4), dim3(256), 0, st, d_pics);
    if (marks) hipEventRecord(marks[3], st);
}

}  // namespace jmamd
