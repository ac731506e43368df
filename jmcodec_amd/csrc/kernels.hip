// jmcodec_amd/csrc/kernels.hip -- gfx950 reconstruction kernels of the jm_amd_dec backend.
//
// Together these kernels are the device half of what the reference delegates to the NVDEC ASIC
// through cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:33-41) plus the host repack of
// jm_nvdec_output_frame (nv_dec.cpp:750-828):
//   k_recon_inter   sub-pel motion compensation + dequant/inverse transform + I_PCM   (fully parallel)
//   k_recon_intra   Intra4x4 / Intra16x16 / chroma intra prediction + residual        (MB wavefront)
//   k_deblock       in-loop deblocking filter, clause 8.7 order                        (MB wavefront)
//   k_packout       pitch NV12 surface -> tight NV12 / I420 display frame              (fully parallel)
// All arithmetic is 8-bit integer pixel work: HBM/LDS bound, no MFMA.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "jobs.h"
#include "kernels.h"
#include "kernel_common.h"
#include "recon_device.h"      // ResTile, mb_residual_to_lds, luma_sample, recon_inter_wave (shared with chain.hip)

namespace jmamd {

// ------------------------------------------------------------------------------------------
// k_recon_inter: one wave per macroblock, 4 macroblocks per workgroup
// ------------------------------------------------------------------------------------------
// Register budget (round 4): the instantiation without the two-list window path must fit 96 registers -- five waves per SIMD.  Round 3's field-picture
// plumbing (parity bit of a reference entry, chroma vector offset) had taken it to 100, i.e. to four waves: +10 % per picture and most of the round's
// 3-5 % loss on the default bench (profiles/r04_ab_r2_vs_r3.json: the round-2 library beside round 3's on one box).  Batches without a field picture --
// all of them in progressive streams -- now run an instantiation compiled without that plumbing (FIELD = false).
template <bool HAS_BI, bool FIELD>
__global__ __launch_bounds__(256) void k_recon_inter(const PicParams *pics, int *err) {
    const PicParams &pp = pics[blockIdx.y];
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so give every XCD one
    // contiguous band of the picture; neighbouring macroblocks, whose reference windows overlap, then share an L2.
    const int per_xcd = ((int)gridDim.x + 7) >> 3;
    const int blk = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (!(pp.stages & PS_RECON) || blk * 4 >= pp.mb_w * pp.mb_h) return;
    __shared__ ReconLds sm;
    const int n_mbs = pp.mb_w * pp.mb_h;
    const int mb = blk * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // wave-uniform, and known to be (recon_device.h)
    recon_inter_wave<false, false, HAS_BI, FIELD>(pp, mb, mb < n_mbs, sm, ChainView{nullptr, err});      // (err: one word per picture of the batch)
}

// ------------------------------------------------------------------------------------------
// wavefront plumbing shared by k_recon_intra and k_deblock: one workgroup of kWaves waves, wave w
// walks macroblock rows w, w+kWaves, ...; progress[row] = number of finished macroblocks in that row.
// A macroblock (x, y) may start once row y-1 has finished min(x+2, mb_w) macroblocks
// (left neighbour is the same wave; top, top-left and top-right are covered by the count).
// ------------------------------------------------------------------------------------------
constexpr int kWaves = 16;
// k_recon_intra runs eight: a workgroup of 16 waves is four per SIMD, i.e. 128 registers each, and the kernel then spilled 63 of them (368 bytes of scratch);
// with two waves per SIMD it keeps everything in registers.  It only sees pictures with a few scattered intra macroblocks (decoder.cpp).
constexpr int kIntraWaves = 8;
constexpr int kMaxRows = 512;

__device__ __forceinline__ void wait_row(volatile int *progress, int row, int need) {
    if (row < 0) return;
    while (progress[row] < need) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void publish_row(volatile int *progress, int row, int done, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) progress[row] = done;
}

// ------------------------------------------------------------------------------------------
// k_recon_intra
// ------------------------------------------------------------------------------------------
struct IntraTile {
    uint8_t y[17][28];        // [0] = row above; column 0 = left neighbour, columns 1..16 MB, 17..24 top-right (4 for Intra4x4, 8 for Intra8x8)
    uint8_t e8[2][32];        // Intra8x8: raw / filtered reference samples, index 0 corner, 1..16 p[0..15,-1], 17..24 p[-1,0..7]
    uint8_t c[2][9][12];      // per plane: [0] = row above, column 0 = left neighbour, 1..8 MB
};

__device__ void intra4x4_block(IntraTile &t, const ResTile &rt, int blk, int mode, bool availA, bool availB, bool availC, bool availD, int lane) {
    // lanes 0..15: pixel (px, py) of the 4x4 block
    if (lane >= 16) return;
    int rpos = blk_to_raster(blk), bx = rpos & 3, by = rpos >> 2;
    int px = lane & 3, py = lane >> 2;
    int ox = 1 + bx * 4, oy = 1 + by * 4;              // tile coordinates of the block origin
    int T[9], L[5];                                    // T[k+1] = p[k,-1] for k=-1..7 ; L[k+1] = p[-1,k] for k=-1..3
#pragma unroll
    for (int k = 0; k < 4; k++) { T[k + 1] = availB ? t.y[oy - 1][ox + k] : 128; L[k + 1] = availA ? t.y[oy + k][ox - 1] : 128; }
#pragma unroll
    for (int k = 4; k < 8; k++) T[k + 1] = (availB && availC) ? t.y[oy - 1][ox + k] : T[4];
    T[0] = L[0] = availD ? t.y[oy - 1][ox - 1] : 128;
#define TT(k) T[(k) + 1]
#define LL(k) L[(k) + 1]
    int p;
    switch (mode) {
    case 0: p = TT(px); break;
    case 1: p = LL(py); break;
    case 2:
        if (availA && availB) p = (TT(0) + TT(1) + TT(2) + TT(3) + LL(0) + LL(1) + LL(2) + LL(3) + 4) >> 3;
        else if (availA) p = (LL(0) + LL(1) + LL(2) + LL(3) + 2) >> 2;
        else if (availB) p = (TT(0) + TT(1) + TT(2) + TT(3) + 2) >> 2;
        else p = 128;
        break;
    case 3: p = (px == 3 && py == 3) ? (TT(6) + 3 * TT(7) + 2) >> 2 : (TT(px + py) + 2 * TT(px + py + 1) + TT(px + py + 2) + 2) >> 2; break;
    case 4:
        if (px > py) p = (TT(px - py - 2) + 2 * TT(px - py - 1) + TT(px - py) + 2) >> 2;
        else if (px < py) p = (LL(py - px - 2) + 2 * LL(py - px - 1) + LL(py - px) + 2) >> 2;
        else p = (TT(0) + 2 * TT(-1) + LL(0) + 2) >> 2;
        break;
    case 5: { int z = 2 * px - py, i = px - (py >> 1);
        if (z >= 0) p = (z & 1) ? (TT(i - 2) + 2 * TT(i - 1) + TT(i) + 2) >> 2 : (TT(i - 1) + TT(i) + 1) >> 1;
        else if (z == -1) p = (LL(0) + 2 * TT(-1) + TT(0) + 2) >> 2;
        else p = (LL(py - 1) + 2 * LL(py - 2) + LL(py - 3) + 2) >> 2;
        break; }
    case 6: { int z = 2 * py - px, i = py - (px >> 1);
        if (z >= 0) p = (z & 1) ? (LL(i - 2) + 2 * LL(i - 1) + LL(i) + 2) >> 2 : (LL(i - 1) + LL(i) + 1) >> 1;
        else if (z == -1) p = (LL(0) + 2 * TT(-1) + TT(0) + 2) >> 2;
        else p = (TT(px - 1) + 2 * TT(px - 2) + TT(px - 3) + 2) >> 2;
        break; }
    case 7: { int i = px + (py >> 1); p = (py & 1) ? (TT(i) + 2 * TT(i + 1) + TT(i + 2) + 2) >> 2 : (TT(i) + TT(i + 1) + 1) >> 1; break; }
    default: { int z = px + 2 * py, i = py + (px >> 1);
        if (z > 5) p = LL(3);
        else if (z == 5) p = (LL(2) + 3 * LL(3) + 2) >> 2;
        else p = (z & 1) ? (LL(i) + 2 * LL(i + 1) + LL(i + 2) + 2) >> 2 : (LL(i) + LL(i + 1) + 1) >> 1;
        break; }
    }
#undef TT
#undef LL
    p = clip1(p + rt.y[(by * 4 + py) * 16 + bx * 4 + px]);
    t.y[oy + py][ox + px] = (uint8_t)p;
}

// Intra8x8 (8.3.2): reference sample filtering 8.3.2.2.1 and the nine modes 8.3.2.2.2-10 for 8x8 block b8; lane = pixel
__device__ void intra8x8_block(IntraTile &t, const ResTile &rt, int b8, int mode, bool availA, bool availB, bool availC, bool availD, int lane) {
    int ox = 1 + (b8 & 1) * 8, oy = 1 + (b8 >> 1) * 8;
    uint8_t *raw = t.e8[0], *fe = t.e8[1];
    if (lane == 0) raw[0] = availD ? t.y[oy - 1][ox - 1] : 128;
    else if (lane <= 16) { int k = lane - 1; raw[lane] = availB ? t.y[oy - 1][ox + ((k >= 8 && !availC) ? 7 : k)] : 128; }
    else if (lane <= 24) raw[lane] = availA ? t.y[oy + lane - 17][ox - 1] : 128;
    // same wave: LDS keeps program order
    if (lane == 0) {
        int c = raw[0], t0 = raw[1], l0 = raw[17];
        fe[0] = (uint8_t)(!availD ? 128 : ((availA &&
            availB) ? (t0 + 2 * c + l0 + 2) >> 2 : (availB ? (3 * c + t0 + 2) >> 2 : (availA ? (3 * c + l0 + 2) >> 2 : c))));
    } else if (lane <= 16) {
        int k = lane - 1, c = raw[lane];
        int lo = k == 0 ? (availD ? raw[0] : c) : raw[lane - 1], hi = k == 15 ? c : raw[lane + 1];
        fe[lane] = (uint8_t)((lo + 2 * c + hi + 2) >> 2);
    } else if (lane <= 24) {
        int k = lane - 17, c = raw[lane];
        int lo = k == 0 ? (availD ? raw[0] : c) : raw[lane - 1], hi = k == 7 ? c : raw[lane + 1];
        fe[lane] = (uint8_t)((lo + 2 * c + hi + 2) >> 2);
    }
    int x = lane & 7, y = lane >> 3;
#define TT(i) ((i) < 0 ? fe[0] : fe[1 + (i)])
#define LL(i) ((i) < 0 ? fe[0] : fe[17 + (i)])
    int p;
    switch (mode) {
    case 0: p = TT(x); break;
    case 1: p = LL(y); break;
    case 2: {
        int st = 0, sl = 0;
        for (int i = 0; i < 8; i++) { st += TT(i); sl += LL(i); }
        p = (availA && availB) ? (st + sl + 8) >> 4 : (availA ? (sl + 4) >> 3 : (availB ? (st + 4) >> 3 : 128));
        break; }
    case 3: p = (x == 7 && y == 7) ? (TT(14) + 3 * TT(15) + 2) >> 2 : (TT(x + y) + 2 * TT(x + y + 1) + TT(x + y + 2) + 2) >> 2; break;
    case 4:
        if (x > y) p = (TT(x - y - 2) + 2 * TT(x - y - 1) + TT(x - y) + 2) >> 2;
        else if (x < y) p = (LL(y - x - 2) + 2 * LL(y - x - 1) + LL(y - x) + 2) >> 2;
        else p = (TT(0) + 2 * TT(-1) + LL(0) + 2) >> 2;
        break;
    case 5: { int z = 2 * x - y, i = x - (y >> 1);
        if (z >= 0) p = (z & 1) ? (TT(i - 2) + 2 * TT(i - 1) + TT(i) + 2) >> 2 : (TT(i - 1) + TT(i) + 1) >> 1;
        else if (z == -1) p = (LL(0) + 2 * TT(-1) + TT(0) + 2) >> 2;
        else p = (LL(y - 2 * x - 1) + 2 * LL(y - 2 * x - 2) + LL(y - 2 * x - 3) + 2) >> 2;
        break; }
    case 6: { int z = 2 * y - x, i = y - (x >> 1);
        if (z >= 0) p = (z & 1) ? (LL(i - 2) + 2 * LL(i - 1) + LL(i) + 2) >> 2 : (LL(i - 1) + LL(i) + 1) >> 1;
        else if (z == -1) p = (LL(0) + 2 * TT(-1) + TT(0) + 2) >> 2;
        else p = (TT(x - 2 * y - 1) + 2 * TT(x - 2 * y - 2) + TT(x - 2 * y - 3) + 2) >> 2;
        break; }
    case 7: { int i = x + (y >> 1); p = (y & 1) ? (TT(i) + 2 * TT(i + 1) + TT(i + 2) + 2) >> 2 : (TT(i) + TT(i + 1) + 1) >> 1; break; }
    default: { int z = x + 2 * y, i = y + (x >> 1);
        if (z > 13) p = LL(7);
        else if (z == 13) p = (LL(6) + 3 * LL(7) + 2) >> 2;
        else p = (z & 1) ? (LL(i) + 2 * LL(i + 1) + LL(i + 2) + 2) >> 2 : (LL(i) + LL(i + 1) + 1) >> 1;
        break; }
    }
#undef TT
#undef LL
    p = clip1(p + rt.y[((b8 >> 1) * 8 + y) * 16 + (b8 & 1) * 8 + x]);
    t.y[oy + y][ox + x] = (uint8_t)p;
}

// plane prediction (8.3.3.4 / 8.3.4.4) for an n x n block whose neighbours sit in a tile with the given row stride
__device__ __forceinline__ int plane_pred(const uint8_t *tile, int stride, int n, int x, int y) {
    // tile points at the MB origin inside the tile (tile[-stride] = row above, tile[-1] = left column)
    int h2 = n >> 1, Hs = 0, Vs = 0;
    for (int k = 0; k < h2; k++) {
        Hs += (k + 1) * (tile[-stride + h2 + k] - tile[-stride + h2 - 2 - k]);
        Vs += (k + 1) * (tile[(h2 + k) * stride - 1] - tile[(h2 - 2 - k) * stride - 1]);
    }
    int a = 16 * (tile[(n - 1) * stride - 1] + tile[-stride + n - 1]);
    int b = n == 16 ? (5 * Hs + 32) >> 6 : (34 * Hs + 32) >> 6;
    int c = n == 16 ? (5 * Vs + 32) >> 6 : (34 * Vs + 32) >> 6;
    return clip1((a + b * (x - (h2 - 1)) + c * (y - (h2 - 1)) + 16) >> 5);
}

__device__ void intra_mb(const PicParams &pp, const MbRec &r, int mbx, int mby, IntraTile &t, ResTile &rt, int lane) {
    int pitch = pp.pitch, W = pp.mb_w * 16, H = pp.mb_h * 16;
    uint8_t *dst = cur_plane(pp);
    uint8_t *dst_c = dst + pp.chroma_offset;
    bool availA = r.flags & MBF_AVAIL_A, availB = r.flags & MBF_AVAIL_B, availC = r.flags & MBF_AVAIL_C, availD = r.flags & MBF_AVAIL_D;
    // ---- residual (zeros when absent) ----
    if (mb_has_residual(r)) mb_residual_to_lds(pp, r, rt, lane);
    else { for (int k = lane; k < 256; k += 64) rt.y[k] = 0; for (int k = lane; k < 128; k += 64) (&rt.c[0][0])[k] = 0; }
    // ---- neighbours into the tile (clamped addresses; unavailable ones are never used) ----
    {
        int x0 = mbx * 16, y0 = mby * 16;
        if (lane < 25) { int x = clip3(0, W - 1, x0 - 1 + lane), y = clip3(0, H - 1, y0 - 1); t.y[0][lane] = dst[(size_t)y * pitch + x]; }
        else if (lane >= 32 && lane < 48) { int i = lane - 32; int x = clip3(0, W - 1, x0 - 1); t.y[1 + i][0] = dst[(size_t)(y0 + i) * pitch + x]; }
        int cx0 = mbx * 8, cy0 = mby * 8, CW = W >> 1, CH = H >> 1;
        if (lane < 18) { int pl = lane / 9, i = lane % 9; int x = clip3(0, CW - 1, cx0 - 1 + i), y = clip3(0, CH - 1, cy0 - 1);
            t.c[pl][0][i] = dst_c[(size_t)y * pitch + 2 * x + pl]; }
        else if (lane >= 32 && lane < 48) { int pl = (lane - 32) >> 3, i = (lane - 32) & 7; int x = clip3(0, CW - 1, cx0 - 1);
            t.c[pl][1 + i][0] = dst_c[(size_t)(cy0 + i) * pitch + 2 * x + pl]; }
    }
    // ---- luma ----
    if (r.kind == MB_I4 && (r.modes & MBM_T8X8)) {
        for (int b8 = 0; b8 < 4; b8++) {
            int bx = b8 & 1, by = b8 >> 1;
            int mode = (r.u.i4[b8 >> 1] >> ((b8 & 1) * 4)) & 15;
            bool a = bx > 0 || availA, b = by > 0 || availB, d = (bx > 0 && by > 0) ? true : (bx > 0 ? availB : (by > 0 ? availA : availD));
            bool c = b8 == 0 ? availB : (b8 == 1 ? availC : b8 == 2);
            intra8x8_block(t, rt, b8, mode, a, b, c, d, lane);
        }
    } else if (r.kind == MB_I4) {
        for (int blk = 0; blk < 16; blk++) {
            int rpos = blk_to_raster(blk), bx = rpos & 3, by = rpos >> 2;
            int mode = (r.u.i4[rpos >> 1] >> ((rpos & 1) * 4)) & 15;
            bool a = bx > 0 || availA, b = by > 0 || availB, d = (bx > 0 && by > 0) ? true : (bx > 0 ? availB : (by > 0 ? availA : availD));
            bool c;
            if (by == 0) c = bx < 3 ? availB : availC;
            else c = !(bx == 3 || blk == 3 || blk == 11 || blk == 7 || blk == 13 || blk == 15);
            intra4x4_block(t, rt, blk, mode, a, b, c, d, lane);
        }
    } else {
        int mode = (r.modes >> 2) & 3;
        int rb = lane >> 2, row = lane & 3, bx = rb & 3, by = rb >> 2, y = by * 4 + row;
        int dc = 128;
        if (mode == 2) {
            int st = 0, sl = 0;
            for (int i = 0; i < 16; i++) { st += t.y[0][1 + i]; sl += t.y[1 + i][0]; }
            dc = (availA && availB) ? (st + sl + 16) >> 5 : (availA ? (sl + 8) >> 4 : (availB ? (st + 8) >> 4 : 128));
        }
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int x = bx * 4 + k, p;
            if (mode == 0) p = t.y[0][1 + x];
            else if (mode == 1) p = t.y[1 + y][0];
            else if (mode == 2) p = dc;
            else p = plane_pred(&t.y[1][1], 28, 16, x, y);
            v[k] = clip1(p + rt.y[y * 16 + x]);
        }
        // all lanes read the borders before anyone overwrites the tile interior (interior is not read for I16)
#pragma unroll
        for (int k = 0; k < 4; k++) t.y[1 + y][1 + bx * 4 + k] = (uint8_t)v[k];
    }
    // ---- chroma: lane -> (cx, cy), both planes ----
    {
        int cmode = r.modes & 3, cx = lane & 7, cy = lane >> 3;
        int out[2];
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            int p;
            if (cmode == 0) {
                int bx = cx >> 2, by = cy >> 2, st = 0, sl = 0;
                for (int i = 0; i < 4; i++) { st += t.c[pl][0][1 + bx * 4 + i]; sl += t.c[pl][1 + by * 4 + i][0]; }
                if (bx == by) p = (availA && availB) ? (st + sl + 4) >> 3 : (availA ? (sl + 2) >> 2 : (availB ? (st + 2) >> 2 : 128));
                else if (bx == 1) p = availB ? (st + 2) >> 2 : (availA ? (sl + 2) >> 2 : 128);
                else p = availA ? (sl + 2) >> 2 : (availB ? (st + 2) >> 2 : 128);
            } else if (cmode == 1) p = t.c[pl][1 + cy][0];
            else if (cmode == 2) p = t.c[pl][0][1 + cx];
            else p = plane_pred(&t.c[pl][1][1], 12, 8, cx, cy);
            out[pl] = clip1(p + rt.c[pl][cy * 8 + cx]);
        }
        *(uint16_t *)(dst_c + (size_t)(mby * 8 + cy) * pitch + mbx * 16 + cx * 2) = (uint16_t)(out[0] | (out[1] << 8));
    }
    // ---- store luma tile ----
    {
        int row = lane >> 2, xq = lane & 3;
        const uint8_t *s = &t.y[1 + row][1 + xq * 4];
        *(uint32_t *)(dst + (size_t)(mby * 16 + row) * pitch + mbx * 16 + xq * 4) = s[0] | (s[1] << 8) | (s[2] << 16) | (s[3] << 24);
    }
}

__global__ __launch_bounds__(kIntraWaves * 64) void k_recon_intra(const PicParams *pics) {
    const PicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & PS_INTRA_V1)) return;
    __shared__ volatile int progress[kMaxRows];
    __shared__ IntraTile tiles[kIntraWaves];
    __shared__ ResTile res[kIntraWaves];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < kMaxRows; i += blockDim.x) progress[i] = 0;
    __syncthreads();
    // progress[row] = x of the first macroblock of the row that is NOT yet reconstructed; inter / I_PCM
    // macroblocks were finished by k_recon_inter, so only Intra4x4 / Intra16x16 ones hold the count back.
    for (int row = wave; row < pp.mb_h; row += kIntraWaves) {
        const MbRec *recs = pp.mbs + (size_t)row * pp.mb_w;
        for (int c0 = 0; c0 < pp.mb_w; c0 += 64) {
            int x = c0 + lane;
            int kind = x < pp.mb_w ? recs[x].kind : MB_INTER;
            unsigned long long m = __ballot(kind == MB_I4 || kind == MB_I16);
            while (m) {
                int xi = c0 + __builtin_ctzll(m);
                m &= m - 1;
                publish_row(progress, row, xi, lane);
                int need = xi + 2 < pp.mb_w ? xi + 2 : pp.mb_w;
                wait_row(progress, row - 1, need);
                MbRec r = recs[xi];
                intra_mb(pp, r, xi, row, tiles[wave], res[wave], lane);
            }
        }
        publish_row(progress, row, pp.mb_w, lane);
    }
}

// ------------------------------------------------------------------------------------------
// k_deblock
// ------------------------------------------------------------------------------------------
// filter one line across an edge; s[0..7] = p3 p2 p1 p0 q0 q1 q2 q3 (luma) in registers
__device__ __forceinline__ void filter_luma(int *s, int bS, int alpha, int beta, const uint8_t *tc0_row) {
    int p3 = s[0], p2 = s[1], p1 = s[2], p0 = s[3], q0 = s[4], q1 = s[5], q2 = s[6], q3 = s[7];
    if (!(iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)) return;
    int ap = iabs(p2 - p0) < beta, aq = iabs(q2 - q0) < beta;
    if (bS < 4) {
        int tc0 = tc0_row[bS - 1], tc = tc0 + ap + aq;
        int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        s[3] = clip1(p0 + delta); s[4] = clip1(q0 - delta);
        if (ap) s[2] = p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1);
        if (aq) s[5] = q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1);
    } else {
        bool strong = iabs(p0 - q0) < ((alpha >> 2) + 2);
        if (ap && strong) { s[3] = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3; s[2] = (p2 + p1 + p0 + q0 + 2) >> 2;
            s[1] = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3; }
        else s[3] = (2 * p1 + p0 + q1 + 2) >> 2;
        if (aq && strong) { s[4] = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3; s[5] = (p0 + q0 + q1 + q2 + 2) >> 2;
            s[6] = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3; }
        else s[4] = (2 * q1 + q0 + p1 + 2) >> 2;
    }
}
// chroma: s[0..3] = p1 p0 q0 q1
__device__ __forceinline__ void filter_chroma(int *s, int bS, int alpha, int beta, const uint8_t *tc0_row) {
    int p1 = s[0], p0 = s[1], q0 = s[2], q1 = s[3];
    if (!(iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)) return;
    if (bS < 4) {
        int tc = tc0_row[bS - 1] + 1;
        int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        s[1] = clip1(p0 + delta); s[2] = clip1(q0 - delta);
    } else { s[1] = (2 * p1 + p0 + q1 + 2) >> 2; s[2] = (2 * q1 + q0 + p1 + 2) >> 2; }
}

struct DbTables { uint8_t alpha[52], beta[52], tc0[52][3]; };   // LDS copy of Tables 8-16 / 8-17
struct DbTile {
    uint8_t y[20][24];      // rows -4..15, cols -4..15 (+pad)
    uint8_t c[10][24];      // interleaved UV: rows -2..7, byte cols -4..15 (chroma cols -2..7)
    uint8_t bs[2][4][4];    // [dir][edge][segment]
};

__device__ void deblock_mb(const PicParams &pp, int mbx, int mby, DbTile &t, const DbTables &tb, int lane) {
    int pitch = pp.pitch, mbw = pp.mb_w;
    uint8_t *dst = cur_plane(pp);
    uint8_t *dst_c = dst + pp.chroma_offset;
    const MbW q = load_mbw(&pp.mbs[mby * mbw + mbx]);
    const SliceRec sl = pp.slices[mbw_slice(q)];
    if (sl.disable == 1) return;
    bool has_left = mbx > 0, has_top = mby > 0;
    MbW pl_ = q, pt_ = q;
    if (has_left) { pl_ = load_mbw(&pp.mbs[mby * mbw + mbx - 1]); if (sl.disable == 2 && mbw_slice(pl_) != mbw_slice(q)) has_left = false; }
    if (has_top) { pt_ = load_mbw(&pp.mbs[(mby - 1) * mbw + mbx]); if (sl.disable == 2 && mbw_slice(pt_) != mbw_slice(q)) has_top = false; }
    // ---- boundary strengths: lanes 0..31 -> (dir, edge, segment) ----
    if (lane < 32) {
        int dir = lane >> 4, e = (lane >> 2) & 3, k = lane & 3;
        int rq = dir == 0 ? k * 4 + e : e * 4 + k;
        int bs;
        if (e == 0) {
            bool have = dir == 0 ? has_left : has_top;
            if (!have) bs = 0;
            else { int rp = dir == 0 ? k * 4 + 3 : 12 + k; bs = boundary_strength(pp, select_mbw(dir == 0, pl_, pt_), rp, q, rq, true, dir == 1); }
        } else bs = boundary_strength(pp, q, dir == 0 ? rq - 1 : rq - 4, q, rq, false, dir == 1);
        if ((e & 1) && (mbw_modes(q) & MBM_T8X8)) bs = 0;    // 8x8 transform: inner 4x4 edges are not filtered
        t.bs[dir][e][k] = (uint8_t)bs;
    }
    // ---- load tiles ----
    int x0 = mbx * 16, y0 = mby * 16;
    {   // luma 20 rows x 5 dwords = 100 dwords; rows above the picture / left of it are never used
        for (int i = lane; i < 100; i += 64) {
            int row = i / 5, dw = i % 5;
            int y = y0 - 4 + row, x = x0 - 4 + dw * 4;
            uint32_t v = 0;
            if (y >= 0 && x >= 0) v = *(const uint32_t *)(dst + (size_t)y * pitch + x);
            *(uint32_t *)&t.y[row][dw * 4] = v;
        }
        for (int i = lane; i < 50; i += 64) {   // chroma 10 rows x 5 dwords
            int row = i / 5, dw = i % 5;
            int y = mby * 8 - 2 + row, x = x0 - 4 + dw * 4;
            uint32_t v = 0;
            if (y >= 0 && x >= 0) v = *(const uint32_t *)(dst_c + (size_t)y * pitch + x);
            *(uint32_t *)&t.c[row][dw * 4] = v;
        }
    }
    // ---- luma vertical edges: lane = pixel row ----
    int qp_q = mbw_qp(q);
    if (lane < 16) {
        for (int e = 0; e < 4; e++) {
            int bs = t.bs[0][e][lane >> 2];
            if (!bs) continue;
            int qp_p = e == 0 ? mbw_qp(pl_) : qp_q;
            int qpav = (qp_p + qp_q + 1) >> 1;
            int ia = clip3(0, 51, qpav + sl.alpha_off), ib = clip3(0, 51, qpav + sl.beta_off);
            int s[8];
            uint8_t *px = &t.y[4 + lane][e * 4];          // p3 is at tile col e*4 (= MB col e*4-4)
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = px[k];
            filter_luma(s, bs, tb.alpha[ia], tb.beta[ib], tb.tc0[ia]);
#pragma unroll
            for (int k = 1; k < 7; k++) px[k] = (uint8_t)s[k];
        }
    }
    // ---- luma horizontal edges: lane = pixel column ----
    if (lane < 16) {
        for (int e = 0; e < 4; e++) {
            int bs = t.bs[1][e][lane >> 2];
            if (!bs) continue;
            int qp_p = e == 0 ? mbw_qp(pt_) : qp_q;
            int qpav = (qp_p + qp_q + 1) >> 1;
            int ia = clip3(0, 51, qpav + sl.alpha_off), ib = clip3(0, 51, qpav + sl.beta_off);
            int s[8];
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = t.y[e * 4 + k][4 + lane];
            filter_luma(s, bs, tb.alpha[ia], tb.beta[ib], tb.tc0[ia]);
#pragma unroll
            for (int k = 1; k < 7; k++) t.y[e * 4 + k][4 + lane] = (uint8_t)s[k];
        }
    }
    // ---- chroma vertical edges (chroma x = 0, 4 <-> luma edges 0, 2): lane = (plane, chroma row) ----
    if (lane < 16) {
        int plane = lane >> 3, row = lane & 7;
        int qc_q = chroma_qp(qp_q, plane ? pp.cr_qp_off : pp.cb_qp_off);
        for (int e = 0; e < 4; e += 2) {
            int bs = t.bs[0][e][row >> 1];
            if (!bs) continue;
            int qc_p = e == 0 ? chroma_qp(mbw_qp(pl_), plane ? pp.cr_qp_off : pp.cb_qp_off) : qc_q;
            int qpav = (qc_p + qc_q + 1) >> 1;
            int ia = clip3(0, 51, qpav + sl.alpha_off), ib = clip3(0, 51, qpav + sl.beta_off);
            int s[4];
            uint8_t *px = &t.c[2 + row][4 + (e * 2) * 2 + plane];     // q0 of this plane at chroma col e*2
#pragma unroll
            for (int k = 0; k < 4; k++) s[k] = px[(k - 2) * 2];
            filter_chroma(s, bs, tb.alpha[ia], tb.beta[ib], tb.tc0[ia]);
            px[-2] = (uint8_t)s[1]; px[0] = (uint8_t)s[2];
        }
    }
    // ---- chroma horizontal edges: lane = interleaved byte column (plane = lane & 1) ----
    if (lane < 16) {
        int plane = lane & 1, ccol = lane >> 1;
        int qc_q = chroma_qp(qp_q, plane ? pp.cr_qp_off : pp.cb_qp_off);
        for (int e = 0; e < 4; e += 2) {
            int bs = t.bs[1][e][ccol >> 1];
            if (!bs) continue;
            int qc_p = e == 0 ? chroma_qp(mbw_qp(pt_), plane ? pp.cr_qp_off : pp.cb_qp_off) : qc_q;
            int qpav = (qc_p + qc_q + 1) >> 1;
            int ia = clip3(0, 51, qpav + sl.alpha_off), ib = clip3(0, 51, qpav + sl.beta_off);
            int s[4];
#pragma unroll
            for (int k = 0; k < 4; k++) s[k] = t.c[e * 2 + k][4 + lane];          // rows (e*2-2 .. e*2+1) + 2
            filter_chroma(s, bs, tb.alpha[ia], tb.beta[ib], tb.tc0[ia]);
            t.c[e * 2 + 1][4 + lane] = (uint8_t)s[1]; t.c[e * 2 + 2][4 + lane] = (uint8_t)s[2];
        }
    }
    // ---- write back (rows -3..15 luma, -1..7 chroma; whole 20-byte rows where they exist) ----
    for (int i = lane; i < 100; i += 64) {
        int row = i / 5, dw = i % 5;
        int y = y0 - 4 + row, x = x0 - 4 + dw * 4;
        if (row >= 1 && y >= 0 && x >= 0) *(uint32_t *)(dst + (size_t)y * pitch + x) = *(const uint32_t *)&t.y[row][dw * 4];
    }
    for (int i = lane; i < 50; i += 64) {
        int row = i / 5, dw = i % 5;
        int y = mby * 8 - 2 + row, x = x0 - 4 + dw * 4;
        if (row >= 1 && y >= 0 && x >= 0) *(uint32_t *)(dst_c + (size_t)y * pitch + x) = *(const uint32_t *)&t.c[row][dw * 4];
    }
}

__global__ __launch_bounds__(kWaves * 64) void k_deblock(const PicParams *pics) {
    const PicParams &pp = pics[blockIdx.y];
    if (!(pp.stages & PS_DEBLOCK_V1)) return;
    __shared__ volatile int progress[kMaxRows];
    __shared__ DbTile tiles[kWaves];
    __shared__ DbTables tb;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < kMaxRows; i += blockDim.x) progress[i] = 0;
    if (threadIdx.x < 52) { tb.alpha[threadIdx.x] = kAlpha[threadIdx.x]; tb.beta[threadIdx.x] = kBeta[threadIdx.x]; }
    if (threadIdx.x < 156) (&tb.tc0[0][0])[threadIdx.x] = (&kTc0[0][0])[threadIdx.x];
    __syncthreads();
    for (int row = wave; row < pp.mb_h; row += kWaves) {
        for (int x = 0; x < pp.mb_w; x++) {
            int need = x + 2 < pp.mb_w ? x + 2 : pp.mb_w;
            wait_row(progress, row - 1, need);
            deblock_mb(pp, x, row, tiles[wave], tb, lane);
            publish_row(progress, row, x + 1, lane);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_packout: restates jm_nvdec_output_frame (nv_dec.cpp:782-820) on the device.
// out_fmt 0: tight NV12; out_fmt 1: Y plane, U plane, V plane ("YV12" in the reference's words, I420 order).
// One thread moves 16 source bytes.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_packout(const PackJob *jobs) {
    const PackJob jb = jobs[blockIdx.y];
    const uint8_t *src = jb.src; uint8_t *dst = jb.dst;
    const int pitch = jb.pitch, chroma_offset = jb.chroma_offset, width = jb.width, height = jb.height, out_fmt = jb.out_fmt;
    int chunks_per_row = (width + 15) >> 4;
    int luma_chunks = chunks_per_row * height;
    int h2 = height >> 1, w2 = width >> 1;
    int total = luma_chunks + chunks_per_row * h2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        bool chroma = i >= luma_chunks;
        int j = chroma ? i - luma_chunks : i;
        int row = j / chunks_per_row, x = (j % chunks_per_row) * 16;
        // (a frame of which only one field was decoded: every row shows the line of that parity of its line pair)
        const int srow = jb.lone_field ? ((row & ~1) | (jb.lone_field - 1)) : row;
        const uint8_t *s = src + (chroma ? chroma_offset : 0) + (size_t)srow * pitch + x;
        int n = width - x < 16 ? width - x : 16;
        if (!chroma || out_fmt == 0) {
            uint8_t *d = dst + (chroma ? (size_t)width * height : 0) + (size_t)row * width + x;
            if (n == 16 && ((((uintptr_t)d) & 15) == 0)) *(uint4 *)d = *(const uint4 *)s;
            else for (int k = 0; k < n; k++) d[k] = s[k];
        } else {
            uint8_t *du = dst + (size_t)width * height + (size_t)row * w2 + (x >> 1);
            uint8_t *dv = du + (size_t)w2 * h2;
            if (n == 16 && ((((uintptr_t)du) & 7) == 0) && ((((uintptr_t)dv) & 7) == 0)) {
                uint4 v = *(const uint4 *)s;
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
                uint32_t u[2], vv[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    uint32_t a = w[2 * k], b = w[2 * k + 1];
                    u[k] = (a & 0xff) | ((a >> 8) & 0xff00) | ((b & 0xff) << 16) | ((b << 8) & 0xff000000u);
                    vv[k] = ((a >> 8) & 0xff) | ((a >> 16) & 0xff00) | ((b << 8) & 0xff0000) | (b & 0xff000000u);
                }
                *(uint2 *)du = make_uint2(u[0], u[1]); *(uint2 *)dv = make_uint2(vv[0], vv[1]);
            } else for (int k = 0; k < n / 2; k++) { du[k] = s[2 * k]; dv[k] = s[2 * k + 1]; }
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers: d_pics / d_jobs are device arrays of n entries; max_* size the grid for the largest picture
// ------------------------------------------------------------------------------------------
void launch_recon_inter(const PicParams *d_pics, int n, int max_mbs, bool any_bipred, bool any_field, int *d_err, hipStream_t st) {
    int nblk = ((max_mbs + 3) / 4 + 7) & ~7;              // multiple of 8: the XCD band mapping must be a bijection
    // any_bipred: some picture of the batch has B slices / weighted prediction (two-list motion records): the instantiation with their LDS-window path
    // any_field: some picture of the batch is a field picture (interlaced streams)
    if (any_field) { if (any_bipred) hipLaunchKernelGGL((k_recon_inter<true, true>), dim3(nblk, n), dim3(256), 0, st, d_pics, d_err);
        else hipLaunchKernelGGL((k_recon_inter<false, true>), dim3(nblk, n), dim3(256), 0, st, d_pics, d_err); }
    else if (any_bipred) hipLaunchKernelGGL((k_recon_inter<true, false>), dim3(nblk, n), dim3(256), 0, st, d_pics, d_err);
    else hipLaunchKernelGGL((k_recon_inter<false, false>), dim3(nblk, n), dim3(256), 0, st, d_pics, d_err);
}
void launch_recon_intra(const PicParams *d_pics, int n, hipStream_t st) { hipLaunchKernelGGL(k_recon_intra, dim3(1, n), dim3(kIntraWaves * 64), 0, st, d_pics);
    }
void launch_deblock(const PicParams *d_pics, int n, hipStream_t st) { hipLaunchKernelGGL(k_deblock, dim3(1, n), dim3(kWaves * 64), 0, st, d_pics); }
// Tight I420 / NV12 frame -> 32-bit ARGB (bytes B, G, R, A), BT.601 limited range, the conversion the reference left behind
// "#if 0" (nv_dec.h:98-107, nv_dec.cpp:244-265).  One thread per pixel pair.
__global__ __launch_bounds__(256) void k_frame_to_argb(const uint8_t *src, int w, int h, int fmt, uint8_t *dst, int dst_pitch) {
    int x2 = (blockIdx.x * 256 + threadIdx.x) * 2, y = blockIdx.y;
    if (x2 >= w || y >= h) return;
    const uint8_t *Y = src + (size_t)y * w + x2;
    int u, v;
    if (fmt == 0) { const uint8_t *c = src + (size_t)w * h + (size_t)(y >> 1) * w + (x2 & ~1); u = c[0]; v = c[1]; }
    else { int cw = w >> 1; const uint8_t *pu = src + (size_t)w * h + (size_t)(y >> 1) * cw + (x2 >> 1); u = pu[0]; v = pu[(size_t)cw * (h >> 1)]; }
    int d = u - 128, e = v - 128;
    uint32_t *o = (uint32_t *)(dst + (size_t)y * dst_pitch) + x2;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (x2 + k >= w) break;
        int c = 298 * (Y[k] - 16) + 128;
        int r = clip1((c + 409 * e) >> 8), g = clip1((c - 100 * d - 208 * e) >> 8), b = clip1((c + 516 * d) >> 8);
        o[k] = 0xFF000000u | ((uint32_t)r << 16) | ((uint32_t)g << 8) | (uint32_t)b;
    }
}
void launch_frame_to_argb(const uint8_t *d_src, int w, int h, int fmt, uint8_t *d_dst, int dst_pitch, hipStream_t st) {
    hipLaunchKernelGGL(k_frame_to_argb, dim3((w / 2 + 255) / 256, h), dim3(256), 0, st, d_src, w, h, fmt, d_dst, dst_pitch);
}

// SURVEY 8f f4 -- the encoder-side pre-processing of the reference as a HIP kernel: tight I420 (or tight NV12) frame -> pitch-linear NV12
// surface, i.e. the cuMemcpy2D of the luma plane plus the "InterleaveUV" kernel of /root/reference/nv_enc/nv_enc.cpp:1022-1079 (arguments
// U, V, dst chroma, chroma width / height, source strides, dst stride) in one launch, device to device.  It is the inverse of k_packout.
// One thread per 4 output bytes of a row (luma rows first, then the h/2 interleaved chroma rows).
__global__ __launch_bounds__(256) void k_frame_to_nv12_pitch(const uint8_t *src, int w, int h, int fmt, uint8_t *dst, int pitch) {
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, row = blockIdx.y;
    if (x >= w) return;
    uint8_t *o = dst + (size_t)row * pitch + x;
    uint8_t v[4];
    if (row < h || fmt == 0) {                                   // luma row, or an already interleaved chroma row: plain copy
        const uint8_t *i = src + (size_t)row * w + x;
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = x + k < w ? i[k] : 0;
    } else {                                                     // chroma row r: bytes 2c, 2c+1 = U[r][c], V[r][c]
        const int cw = w >> 1, r = row - h;
        const uint8_t *pu = src + (size_t)w * h + (size_t)r * cw, *pv = pu + (size_t)cw * (h >> 1);
#pragma unroll
        for (int k = 0; k < 4; k++) { const int c = (x + k) >> 1; v[k] = x + k < w ? (((x + k) & 1) ? pv[c] : pu[c]) : 0; }
    }
    if (x + 4 <= w && !(pitch & 3)) *(uint32_t *)o = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
    else for (int k = 0; k < 4 && x + k < w; k++) o[k] = v[k];
}
void launch_frame_to_nv12_pitch(const uint8_t *d_src, int w, int h, int fmt, uint8_t *d_dst, int pitch, hipStream_t st) {
    hipLaunchKernelGGL(k_frame_to_nv12_pitch, dim3(((w + 3) / 4 + 255) / 256, h + h / 2), dim3(256), 0, st, d_src, w, h, fmt, d_dst, pitch);
}

void launch_packout(const PackJob *d_jobs, int n, int max_width, int max_height, hipStream_t st) {
    int chunks = ((max_width + 15) >> 4) * (max_height + (max_height >> 1));
    int blocks = (chunks + 255) / 256;
    // The destination is pinned HOST memory: the kernel is PCIe-bound (~55 GB/s), not CU-bound.  A small grid is enough to
    // keep the link full and leaves the CUs to the decode kernels of the next batch that run concurrently.
    const int total = 160;
    int cap = total / (n > 0 ? n : 1);
    if (cap < 1) cap = 1;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_packout, dim3(blocks, n), dim3(256), 0, st, d_jobs);
}

}  // namespace jmamd
