// jmcodec_amd/csrc/hevc_kernels.hip -- HEVC sample reconstruction on gfx950 (ITU-T H.265 8.4.4.2, 8.5.3.3, 8.6.4.2, 8.7.2.5, 8.7.3).
//
// The device half of the replacement for cuvidDecodePicture with codec_type 1 (/root/reference/nv_dec/nv_dec.cpp:33-41): the host
// (hevc_slice.cpp) delivers motion-compensation blocks, scaled coefficients, intra blocks with their neighbour availability,
// boundary strengths and SAO parameters (hevc_jobs.h); these kernels produce the samples.  One launch serves a batch of pictures
// (blockIdx.y = picture), as for H.264.
//   k_hevc_mc        one wavefront per block of <= 16x16 luma samples: 8-tap / 4-tap separable interpolation through LDS, weighting
//   k_hevc_resid     one wavefront per transform block of an inter CU: sparse coefficients -> inverse transform -> add
//   k_hevc_intra     one wavefront per coding tree block of a wavefront diagonal (x + 2y = d): its intra blocks in decoding order
//   k_hevc_deblock   one lane per 4-sample edge segment; all vertical edges of the picture, then all horizontal edges (8.7.2)
//   k_hevc_sao       one lane per sample, deblocked surface -> final surface (8.7.3)
// All integer arithmetic on 8-bit samples; surfaces are pitch-linear NV12 like the H.264 path's.
#include <hip/hip_runtime.h>
#include "hevc_jobs.h"
#include "hevc_kernels.h"
#include "hevc_tables.h"

namespace jmamd {

__constant__ int8_t c_trans[32][32];
__constant__ int8_t c_dst[4][4];
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
// 14-bit intermediate prediction of one component block: out[i] for the lane's samples k = lane + 64 * i
template <int NT>
__device__ void interp_block(const uint8_t *plane, int pitch, int step, int pw, int ph, int xi, int yi, int bw, int bh, int xf, int yf,
                             const int8_t *fx, const int8_t *fy, uint8_t *tile, int16_t *hbuf, int lane, int *out) {
    const int tw = bw + NT - 1, th = bh + NT - 1, off = NT / 2 - 1;
    __syncthreads();
    for (int k = lane; k < tw * th; k += 64) {
        const int r = k / tw, c = k - r * tw;
        tile[k] = plane[(size_t)clip3(0, ph - 1, yi + r - off) * pitch + clip3(0, pw - 1, xi + c - off) * step];
    }
    __syncthreads();
    for (int k = lane; k < bw * th; k += 64) {
        const int r = k / bw, c = k - r * bw;
        int v;
        if (xf) { v = 0; for (int i = 0; i < NT; i++) v += fx[i] * tile[r * tw + c + i]; } else v = tile[r * tw + c + off];
        hbuf[k] = (int16_t)v;
    }
    __syncthreads();
    for (int i = 0; i < 4; i++) {
        const int k = lane + 64 * i;
        if (k >= bw * bh) break;
        const int r = k / bw, c = k - r * bw;
        int v;
        if (yf) { v = 0; for (int j = 0; j < NT; j++) v += fy[j] * hbuf[(r + j) * bw + c]; if (xf) v >>= 6; }
        else { v = hbuf[(r + off) * bw + c]; if (!xf) v <<= 6; }
        out[i] = v;
    }
}

__global__ __launch_bounds__(64) void k_hevc_mc(const HevcPicParams *pics) {
    const HevcPicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & HPS_MC) || (int)blockIdx.x >= pp.n_pus) return;
    const HevcPu pu = pp.pus[blockIdx.x];
    const int lane = threadIdx.x;
    __shared__ uint8_t tile[23 * 23 + 3];
    __shared__ int16_t hbuf[23 * 16];
    uint8_t *dst = pp.surf[pp.work];
    const HevcWp *wp = pu.wp ? &pp.wps[pu.wp - 1] : nullptr;
    const int both = pu.slot0 >= 0 && pu.slot1 >= 0;
    for (int c = 0; c < 3; c++) {
        const int sc = c ? 1 : 0, bw = pu.w >> sc, bh = pu.h >> sc, xb = pu.x >> sc, yb = pu.y >> sc, pw = pp.w >> sc, ph = pp.h >> sc;
        int p[2][4];
        for (int l = 0; l < 2; l++) {
            const int slot = l ? pu.slot1 : pu.slot0;
            if (slot < 0) continue;
            const int mvx = l ? pu.mv1[0] : pu.mv0[0], mvy = l ? pu.mv1[1] : pu.mv0[1];
            const uint8_t *ref = pp.surf[slot];
            if (c == 0) interp_block<8>(ref, pp.pitch, 1, pw, ph, xb + (mvx >> 2), yb + (mvy >> 2), bw, bh, mvx & 3, mvy & 3, c_lf[mvx & 3], c_lf[mvy & 3], tile, hbuf, lane, p[l]);
            else interp_block<4>(ref + pp.chroma_offset + (c - 1), pp.pitch, 2, pw, ph, xb + (mvx >> 3), yb + (mvy >> 3), bw, bh, mvx & 7, mvy & 7, c_cf[mvx & 7], c_cf[mvy & 7], tile, hbuf, lane, p[l]);
        }
        for (int i = 0; i < 4; i++) {
            const int k = lane + 64 * i;
            if (k >= bw * bh) break;
            const int r = k / bw, cc = k - r * bw;
            int v;
            if (!wp) v = both ? (p[0][i] + p[1][i] + 64) >> 7 : ((pu.slot0 >= 0 ? p[0][i] : p[1][i]) + 32) >> 6;          // 8.5.3.3.4.2
            else {                                                                                                     // 8.5.3.3.4.3
                const int ld = wp->log2wd[c ? 1 : 0];
                if (both) v = (p[0][i] * wp->w[0][pu.ridx0][c] + p[1][i] * wp->w[1][pu.ridx1][c] + ((wp->o[0][pu.ridx0][c] + wp->o[1][pu.ridx1][c] + 1) << ld)) >> (ld + 1);
                else if (pu.slot0 >= 0) v = ((p[0][i] * wp->w[0][pu.ridx0][c] + (1 << (ld - 1))) >> ld) + wp->o[0][pu.ridx0][c];
                else v = ((p[1][i] * wp->w[1][pu.ridx1][c] + (1 << (ld - 1))) >> ld) + wp->o[1][pu.ridx1][c];
            }
            *sample_ptr(dst, pp, c, xb + cc, yb + r) = (uint8_t)clip1(v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// 8.6.4.2: residual of one transform block from its sparse scaled coefficients; result in res[y * n + x]
// ------------------------------------------------------------------------------------------------------------
__device__ void residual_block(const uint32_t *coefs, int count, int log2, int flags, int16_t *d, int16_t *res, int lane) {
    const int n = 1 << log2, nn = n * n;
    __syncthreads();
    for (int k = lane; k < nn; k += 64) d[k] = 0;
    __syncthreads();
    for (int k = lane; k < count; k += 64) { const uint32_t e = coefs[k]; d[e & 1023] = (int16_t)(e >> 16); }
    __syncthreads();
    if (flags & HTB_BYPASS) { for (int k = lane; k < nn; k += 64) res[k] = d[k]; __syncthreads(); return; }
    if (flags & HTB_TSKIP) { for (int k = lane; k < nn; k += 64) res[k] = (int16_t)(((d[k] << 7) + 2048) >> 12); __syncthreads(); return; }
    const int step = 32 >> log2, dst = flags & HTB_DST;
    // columns: g[y][x] = clip16((sum_k M[k][y] * d[k][x] + 64) >> 7), kept in res
    for (int k = lane; k < nn; k += 64) {
        const int y = k >> log2, x = k & (n - 1);
        int v = 0;
        for (int j = 0; j < n; j++) { const int dj = d[j * n + x]; if (dj) v += (dst ? c_dst[j][y] : c_trans[j * step][y]) * dj; }
        res[k] = (int16_t)clip3(-32768, 32767, (v + 64) >> 7);
    }
    __syncthreads();
    // rows: r[y][x] = (sum_k M[k][x] * g[y][k] + 2048) >> 12, back into d, then copied to res
    for (int k = lane; k < nn; k += 64) {
        const int y = k >> log2, x = k & (n - 1);
        int v = 0;
        for (int j = 0; j < n; j++) v += (dst ? c_dst[j][x] : c_trans[j * step][x]) * res[y * n + j];
        d[k] = (int16_t)((v + 2048) >> 12);
    }
    __syncthreads();
    for (int k = lane; k < nn; k += 64) res[k] = d[k];
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_hevc_resid(const HevcPicParams *pics) {
    const HevcPicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & HPS_RESID) || (int)blockIdx.x >= pp.n_tbs) return;
    const HevcTb tb = pp.tbs[blockIdx.x];
    __shared__ int16_t d[32 * 32], res[32 * 32];
    const int lane = threadIdx.x, n = 1 << tb.log2;
    residual_block(pp.coefs + tb.coef_off, (int)tb.coef_n, tb.log2, tb.flags, d, res, lane);
    uint8_t *dst = pp.surf[pp.work];
    for (int k = lane; k < n * n; k += 64) {
        const int y = k >> tb.log2, x = k & (n - 1);
        uint8_t *p = sample_ptr(dst, pp, tb.plane, tb.x + x, tb.y + y);
        *p = (uint8_t)clip1(*p + res[k]);
    }
}

// ------------------------------------------------------------------------------------------------------------
// 8.4.4.2: intra prediction (+ residual) of the intra blocks of one coding tree block, in decoding order
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int load_recon(const uint8_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // written by this workgroup moments ago

__global__ __launch_bounds__(64) void k_hevc_intra(const HevcPicParams *pics, int diag) {
    const HevcPicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & HPS_INTRA)) return;
    int ymin = diag - (pp.ctb_w - 1); ymin = ymin > 0 ? (ymin + 1) >> 1 : 0;
    const int cy = ymin + (int)blockIdx.x, cx = diag - 2 * cy;
    if (cy >= pp.ctb_h || cx < 0 || cx >= pp.ctb_w) return;
    const HevcCtb ctb = pp.ctbs[cy * pp.ctb_w + cx];
    if (!ctb.intra_count) return;
    __shared__ int16_t edge[2][132];           // [0] raw, [1] filtered: 0 .. 2n-1 left column bottom-to-top, 2n corner, 2n+1 .. 4n top row left-to-right
    __shared__ uint8_t ok[132];
    __shared__ int16_t refa[32 * 3 + 8];       // main reference of the angular modes, index 0 at refa[32]
    __shared__ int16_t d[32 * 32], res[32 * 32];
    __shared__ int s_first;
    const int lane = threadIdx.x;
    uint8_t *surf = pp.surf[pp.work];
    for (uint32_t ti = 0; ti < ctb.intra_count; ti++) {
        const HevcIntraTb tb = pp.itbs[ctb.intra_first + ti];
        const int log2 = tb.log2, n = 1 << log2, c = tb.plane, N = 4 * n, unit = c ? 2 : 4;
        const bool pcm = tb.mode == kHevcModePcm;
        if (tb.coef_n) residual_block(pp.coefs + tb.coef_off, (int)tb.coef_n, log2, tb.flags, d, res, lane); else __syncthreads();
        const int16_t *e = edge[0];
        if (!pcm) {
            // ---- neighbouring samples (8.4.4.2.2) ----
            for (int i = lane; i <= N; i += 64) {
                int a, v = 0;
                if (i < 2 * n) { const int row = 2 * n - 1 - i; a = (tb.avail >> (row / unit)) & 1; if (a) v = load_recon(sample_ptr(surf, pp, c, tb.x - 1, tb.y + row)); }
                else if (i == 2 * n) { a = (tb.flags & HTB_CORNER) != 0; if (a) v = load_recon(sample_ptr(surf, pp, c, tb.x - 1, tb.y - 1)); }
                else { const int col = i - 2 * n - 1; a = (tb.avail >> (16 + col / unit)) & 1; if (a) v = load_recon(sample_ptr(surf, pp, c, tb.x + col, tb.y - 1)); }
                ok[i] = (uint8_t)a; edge[0][i] = (int16_t)v;
            }
            __syncthreads();
            if (lane == 0) { int f = -1; for (int i = 0; i <= N; i++) if (ok[i]) { f = i; break; } s_first = f; }
            __syncthreads();
            const int first = s_first;
            if (first < 0) { for (int i = lane; i <= N; i += 64) edge[0][i] = 128; }
            else if (lane == 0) { for (int i = 0; i < first; i++) edge[0][i] = edge[0][first]; for (int i = first + 1; i <= N; i++) if (!ok[i]) edge[0][i] = edge[0][i - 1]; }
            __syncthreads();
            // ---- filtering (8.4.4.2.3) ----
            if (c == 0 && tb.mode != 1 && n > 4) {
                const int dv = iabs(tb.mode - 26), dh = iabs(tb.mode - 10), md = dv < dh ? dv : dh, thr = n == 8 ? 7 : (n == 16 ? 1 : 0);
                if (md > thr) {
                    const bool strong = pp.strong_intra && n == 32 && iabs(edge[0][64] + edge[0][128] - 2 * edge[0][96]) < 8 && iabs(edge[0][64] + edge[0][0] - 2 * edge[0][32]) < 8;
                    for (int i = lane; i <= N; i += 64) {
                        int v;
                        if (i == 0 || i == N) v = edge[0][i];
                        else if (strong) v = i == 64 ? edge[0][64] : (i < 64 ? (i * edge[0][64] + (64 - i) * edge[0][0] + 32) >> 6 : ((128 - i) * edge[0][64] + (i - 64) * edge[0][128] + 32) >> 6);
                        else v = (edge[0][i - 1] + 2 * edge[0][i] + edge[0][i + 1] + 2) >> 2;
                        edge[1][i] = (int16_t)v;
                    }
                    e = edge[1];
                }
            }
            __syncthreads();
        }
        const int16_t *L = e + 2 * n - 1, *T = e + 2 * n + 1;      // L[-y] = left sample of row y, T[x] = top sample of column x, T[-1] = corner
        int ang = 0; bool vert = false;
        int dc = 0;
        if (!pcm && tb.mode == 1) { int s = n; for (int i = 0; i < n; i++) s += L[-i] + T[i]; dc = s >> (log2 + 1); }
        if (!pcm && tb.mode >= 2) {
            ang = c_angle[tb.mode]; vert = tb.mode >= 18;
            const int inv = c_inv_angle[tb.mode], lo = ang < 0 ? (n * ang) >> 5 : 0;
            for (int i = lo + lane; i <= 2 * n; i += 64) {
                int v;
                if (i >= 0) v = vert ? T[i - 1] : L[-(i - 1)];
                else { const int k = (i * inv + 128) >> 8; v = vert ? L[-(k - 1)] : T[k - 1]; }
                refa[32 + i] = (int16_t)v;
            }
            __syncthreads();
        }
        const int16_t *ref = refa + 32;
        for (int k = lane; k < n * n; k += 64) {
            const int y = k >> log2, x = k & (n - 1);
            int v;
            if (pcm) v = 0;
            else if (tb.mode == 0) v = ((n - 1 - x) * L[-y] + (x + 1) * T[n] + (n - 1 - y) * T[x] + (y + 1) * L[-n] + n) >> (log2 + 1);
            else if (tb.mode == 1) {
                v = dc;
                if (c == 0 && n < 32) { if (x == 0 && y == 0) v = (L[0] + 2 * dc + T[0] + 2) >> 2; else if (y == 0) v = (T[x] + 3 * dc + 2) >> 2; else if (x == 0) v = (L[-y] + 3 * dc + 2) >> 2; }
            } else {
                const int a = vert ? y : x, b = vert ? x : y, pos = (a + 1) * ang, idx = pos >> 5, fr = pos & 31;
                v = fr ? ((32 - fr) * ref[b + idx + 1] + fr * ref[b + idx + 2] + 16) >> 5 : ref[b + idx + 1];
                if (c == 0 && n < 32 && ang == 0 && b == 0) v = clip1((vert ? T[0] : L[0]) + (((vert ? L[-a] : T[a]) - T[-1]) >> 1));
            }
            if (tb.coef_n) v = clip1(v + res[k]);
            *sample_ptr(surf, pp, c, tb.x + x, tb.y + y) = (uint8_t)v;
        }
        __threadfence();
        __syncthreads();
    }
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
    uint8_t *surf = pp.surf[pp.work];
    {
        const int beta = c_beta[clip3(0, 51, qpl + 2 * cq.beta_off)], tc = c_tc[clip3(0, 53, qpl + 2 * (bs - 1) + 2 * cq.tc_off)];
        const int across = dir ? pp.pitch : 1, along = dir ? 1 : pp.pitch;
        uint8_t *q = surf + (size_t)y * pp.pitch + x;
        int s[4][8];                                                    // s[line][0..7] = p3 p2 p1 p0 q0 q1 q2 q3
        for (int k = 0; k < 4; k++) for (int i = 0; i < 8; i++) s[k][i] = q[k * along + (i - 4) * across];
        const int dp0 = iabs(s[0][1] - 2 * s[0][2] + s[0][3]), dp3 = iabs(s[3][1] - 2 * s[3][2] + s[3][3]), dq0 = iabs(s[0][6] - 2 * s[0][5] + s[0][4]), dq3 = iabs(s[3][6] - 2 * s[3][5] + s[3][4]);
        if (dp0 + dq0 + dp3 + dq3 < beta) {
            bool strong = true;
            for (int k = 0; k < 4; k += 3) { const int dk = k ? dp3 + dq3 : dp0 + dq0; if (!(2 * dk < (beta >> 2) && iabs(s[k][0] - s[k][3]) + iabs(s[k][4] - s[k][7]) < (beta >> 3) && iabs(s[k][3] - s[k][4]) < ((5 * tc + 1) >> 1))) strong = false; }
            const int side = (beta + (beta >> 1)) >> 3; const bool mp = dp0 + dp3 < side, mq = dq0 + dq3 < side;
            for (int k = 0; k < 4; k++) {
                const int p3 = s[k][0], p2 = s[k][1], p1 = s[k][2], p0 = s[k][3], q0 = s[k][4], q1 = s[k][5], q2 = s[k][6], q3 = s[k][7];
                uint8_t *l = q + k * along;
                if (strong) {
                    if (!keep_p) { l[-across] = (uint8_t)clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3); l[-2 * across] = (uint8_t)clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2); l[-3 * across] = (uint8_t)clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3); }
                    if (!keep_q) { l[0] = (uint8_t)clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3); l[across] = (uint8_t)clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2); l[2 * across] = (uint8_t)clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3); }
                } else {
                    int dl = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
                    if (iabs(dl) >= 10 * tc) continue;
                    dl = clip3(-tc, tc, dl);
                    if (!keep_p) { l[-across] = (uint8_t)clip1(p0 + dl); if (mp) l[-2 * across] = (uint8_t)clip1(p1 + clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + dl) >> 1)); }
                    if (!keep_q) { l[0] = (uint8_t)clip1(q0 - dl); if (mq) l[across] = (uint8_t)clip1(q1 + clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - dl) >> 1)); }
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
                uint8_t *p1 = sample_ptr(surf, pp, c, dir ? xc + k : xc - 2, dir ? yc - 2 : yc + k), *p0 = sample_ptr(surf, pp, c, dir ? xc + k : xc - 1, dir ? yc - 1 : yc + k);
                uint8_t *q0 = sample_ptr(surf, pp, c, dir ? xc + k : xc, dir ? yc : yc + k), *q1 = sample_ptr(surf, pp, c, dir ? xc + k : xc + 1, dir ? yc + 1 : yc + k);
                const int dl = clip3(-tc, tc, (((*q0 - *p0) << 2) + *p1 - *q1 + 4) >> 3), np = clip1(*p0 + dl), nq = clip1(*q0 - dl);
                if (!keep_p) *p0 = (uint8_t)np;
                if (!keep_q) *q0 = (uint8_t)nq;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// 8.7.3: sample adaptive offset, work surface -> current surface.  One lane per sample; blockIdx.z = plane
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hevc_sao(const HevcPicParams *pics) {
    const HevcPicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & HPS_SAO)) return;
    const int c = blockIdx.z, sc = c ? 1 : 0, pw = pp.w >> sc, ph = pp.h >> sc;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= pw * ph) return;
    const int x = idx % pw, y = idx / pw, xl = x << sc, yl = y << sc;
    const uint8_t *src = pp.surf[pp.work]; uint8_t *dst = pp.surf[pp.cur];
    const int v = *sample_ptr((uint8_t *)src, pp, c, x, y);
    const int cxb = xl >> pp.ctb_log2, cyb = yl >> pp.ctb_log2;
    const HevcCtb &ctb = pp.ctbs[cyb * pp.ctb_w + cxb];
    int add = 0;
    const int type = ctb.sao_type[c];
    if (type && !(pp.qp8[(yl >> 3) * pp.w8 + (xl >> 3)] & 128)) {
        if (type == 1) { const int k = ((v >> 3) - ctb.sao_pos[c]) & 31; if (k < 4) add = ctb.sao_off[c][k]; }
        else {
            const int cls = ctb.sao_pos[c], dx = cls == 1 ? 0 : (cls == 3 ? -1 : 1), dy = cls == 0 ? 0 : 1;      // second neighbour; the first is its mirror image
            const int xa = x - dx, ya = y - dy, xb = x + dx, yb = y + dy;
            bool okk = xa >= 0 && xb >= 0 && xa < pw && xb < pw && ya >= 0 && yb < ph;
            if (okk) {
                // neighbours in another CTB: allowed only where the host's slice / tile analysis says so
                for (int k = 0; k < 2; k++) {
                    const int xn = (k ? xb : xa) << sc, yn = (k ? yb : ya) << sc, ddx = (xn >> pp.ctb_log2) - cxb, ddy = (yn >> pp.ctb_log2) - cyb;
                    if (ddx == 0 && ddy == 0) continue;
                    const int dirk = ddy == 0 ? (ddx < 0 ? 0 : 1) : (ddx == 0 ? (ddy < 0 ? 2 : 3) : (ddy < 0 ? (ddx < 0 ? 4 : 5) : (ddx < 0 ? 6 : 7)));
                    if (!((ctb.nb_mask >> dirk) & 1)) okk = false;
                }
            }
            if (okk) {
                const int a = *sample_ptr((uint8_t *)src, pp, c, xa, ya), b = *sample_ptr((uint8_t *)src, pp, c, xb, yb);
                const int sg = (v > a) - (v < a) + (v > b) - (v < b);
                add = sg == -2 ? ctb.sao_off[c][0] : sg == -1 ? ctb.sao_off[c][1] : sg == 1 ? ctb.sao_off[c][2] : sg == 2 ? ctb.sao_off[c][3] : 0;
            }
        }
    }
    *sample_ptr(dst, pp, c, x, y) = (uint8_t)clip1(v + add);
}

// ------------------------------------------------------------------------------------------------------------
static void upload_tables() {
    static bool done[64] = {false};
    int dev = 0; hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || done[dev]) return;
    hipMemcpyToSymbol(HIP_SYMBOL(c_trans), hevc_trans, sizeof c_trans); hipMemcpyToSymbol(HIP_SYMBOL(c_dst), hevc_dst, sizeof c_dst);
    hipMemcpyToSymbol(HIP_SYMBOL(c_lf), hevc_luma_filter, sizeof c_lf); hipMemcpyToSymbol(HIP_SYMBOL(c_cf), hevc_chroma_filter, sizeof c_cf);
    hipMemcpyToSymbol(HIP_SYMBOL(c_angle), hevc_intra_angle, sizeof c_angle); hipMemcpyToSymbol(HIP_SYMBOL(c_inv_angle), hevc_inv_angle, sizeof c_inv_angle);
    hipMemcpyToSymbol(HIP_SYMBOL(c_beta), hevc_beta_tab, sizeof c_beta); hipMemcpyToSymbol(HIP_SYMBOL(c_tc), hevc_tc_tab, sizeof c_tc); hipMemcpyToSymbol(HIP_SYMBOL(c_qpc), hevc_qpc_tab, sizeof c_qpc);
    done[dev] = true;
}
void hevc_kernels_init() { upload_tables(); }

void launch_hevc_picture_batch(const HevcPicParams *d_pics, int n, const HevcBatchDims &m, hipStream_t st, hipEvent_t *marks) {
    upload_tables();
    if (marks) hipEventRecord(marks[0], st);
    if (m.max_pus > 0) hipLaunchKernelGGL(k_hevc_mc, dim3(m.max_pus, n), dim3(64), 0, st, d_pics);
    if (m.max_tbs > 0) hipLaunchKernelGGL(k_hevc_resid, dim3(m.max_tbs, n), dim3(64), 0, st, d_pics);
    if (marks) hipEventRecord(marks[1], st);
    if (m.any_intra) {
        const int n_diag = m.max_ctb_w + 2 * (m.max_ctb_h - 1), per = m.max_ctb_w < (m.max_ctb_h * 2) ? (m.max_ctb_w + 1) / 2 + 1 : m.max_ctb_h;
        for (int d = 0; d < n_diag; d++) hipLaunchKernelGGL(k_hevc_intra, dim3(per > m.max_ctb_h ? m.max_ctb_h : per, n), dim3(64), 0, st, d_pics, d);
    }
    if (marks) hipEventRecord(marks[2], st);
    if (m.any_deblock) {
        hipLaunchKernelGGL(k_hevc_deblock, dim3(((m.max_w >> 3) * (m.max_h >> 2) + 255) / 256, n), dim3(256), 0, st, d_pics, 0);
        hipLaunchKernelGGL(k_hevc_deblock, dim3(((m.max_w >> 2) * (m.max_h >> 3) + 255) / 256, n), dim3(256), 0, st, d_pics, 1);
    }
    if (m.any_sao) hipLaunchKernelGGL(k_hevc_sao, dim3((m.max_w * m.max_h + 255) / 256, n, 3), dim3(256), 0, st, d_pics);
    if (marks) hipEventRecord(marks[3], st);
}

}  // namespace jmamd
