// jmcodec_amd/csrc/kernel_common.h -- device helpers shared by kernels.hip and deblock_lds.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "jobs.h"

namespace jmamd {

// Field pictures (PicParams.field != 0): the picture is the lines of one parity of its surface -- pitch is twice the surface's, and a reference entry
// (MbRec.ref, the slot1 bytes of a motion record) carries the parity of the reference FIELD in bit 5.  In a frame picture both reduce to surf[slot].
__device__ __forceinline__ const uint8_t *ref_plane(const PicParams &pp, int slot) { return pp.surf[slot & 31] + ((slot & 32) ? (pp.pitch >> 1) : 0); }
__device__ __forceinline__ uint8_t *cur_plane(const PicParams &pp) { return pp.surf[pp.cur] + (pp.field == 2 ? (pp.pitch >> 1) : 0); }
// 8.4.1.4, Table 8-9: the vertical chroma vector of a field that predicts from a field of the other parity: -2 (top from bottom) / +2 (bottom from top)
__device__ __forceinline__ int chroma_mvy_offset(const PicParams &pp, int slot) {
    return pp.field == 0 || ((slot >> 5) & 1) == pp.field - 1 ? 0 : (pp.field == 2 ? 2 : -2);
}

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int clip1(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }

// Small tables are packed into 64-bit immediates so that a lookup is a shift, not a dependent memory
// access inside the serial wavefront chains.
// normAdjust4x4 (8.5.9) by qP%6: class 0 (even,even) {10,11,13,14,16,18}, class 1 (odd,odd) {16,18,20,23,25,29}, class 2 {13,14,16,18,20,23}
__device__ __forceinline__ int norm4(int qp_rem, int cls) {
    unsigned long long t = cls == 0 ? 0x12100E0D0B0AULL : (cls == 1 ? 0x1D1917141210ULL : 0x171412100E0DULL);
    return (int)((t >> (8 * qp_rem)) & 0xff);
}
// Table 8-15: QPc for qPI = 30..51 : 29,30,31,32,32,33,34,34,35,35,36,36,37,37,37,38,38,38,39,39,39,39 (stored minus 29, 4 bits each)
__device__ __forceinline__ int qpc_from_qpi(int qpi) {
    if (qpi < 30) return qpi;
    int k = qpi - 30;
    unsigned long long lo = 0x9888776655433210ULL;                   // entries 0..15, 4 bits each
    unsigned int hi = 0xAAAA99u;                                     // entries 16..21
    int v = k < 16 ? (int)((lo >> (4 * k)) & 15) : (int)((hi >> (4 * (k - 16))) & 15);
    return 29 + v;
}
static __device__ const uint8_t kAlpha[52] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,4,4,5,6,7,8,9,10,12,13,15,17,20,22,25,28,
    32,36,40,45,50,56,63,71,80,90,101,113,127,144,162,182,203,226,255,255 };
static __device__ const uint8_t kBeta[52] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,2,2,3,3,3,3,4,4,4,6,6,7,7,8,8,
    9,9,10,10,11,11,12,12,13,13,14,14,15,15,16,16,17,17,18,18 };
static __device__ const uint8_t kTc0[52][3] = {
 {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},
 {0,0,1},{0,0,1},{0,0,1},{0,0,1},{0,1,1},{0,1,1},{1,1,1},{1,1,1},{1,1,1},{1,1,1},{1,1,2},{1,1,2},{1,1,2},{1,1,2},{1,2,3},{1,2,3},
 {2,2,3},{2,2,4},{2,3,4},{2,3,4},{3,3,5},{3,4,6},{3,4,6},{4,5,7},{4,5,8},{4,6,9},{5,7,10},{6,8,11},{6,8,13},{7,10,14},{8,11,16},
 {9,12,18},{10,13,20},{11,15,23},{13,17,25} };

__device__ __forceinline__ int chroma_qp(int qpy, int off) {
    return qpc_from_qpi(clip3(0, 51, qpy + off));
}
// LevelScale4x4 with the flat (16) weight matrix: 16 * normAdjust4x4 (8.5.9)
__device__ __forceinline__ int level_scale4(int qp_rem, int pos) {
    int i = pos >> 2, j = pos & 3;
    int cls = (!(i & 1) && !(j & 1)) ? 0 : (((i & 1) && (j & 1)) ? 1 : 2);
    return 16 * norm4(qp_rem, cls);
}
// 8.5.12.1 with a weight w from a scaling matrix: LevelScale4x4 = w * normAdjust4x4
__device__ __forceinline__ int dequant4w(int c, int qp, int pos, int w) {
    int i = pos >> 2, j = pos & 3;
    int cls = (!(i & 1) && !(j & 1)) ? 0 : (((i & 1) && (j & 1)) ? 1 : 2);
    int ls = w * norm4(qp % 6, cls), s = qp / 6;
    return s >= 4 ? (c * ls) << (s - 4) : (c * ls + (1 << (3 - s))) >> (4 - s);
}
// 8.5.12.1 scaling of one residual coefficient (not the separately handled DC ones)
__device__ __forceinline__ int dequant4(int c, int qp, int pos) {
    int ls = level_scale4(qp % 6, pos), s = qp / 6;
    return s >= 4 ? (c * ls) << (s - 4) : (c * ls + (1 << (3 - s))) >> (4 - s);
}
// 8.5.12.2 inverse 4x4 transform, in place on d[16] (raster), result already >> 6
__device__ __forceinline__ void idct4x4(int *d) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int a = d[4 * i], b = d[4 * i + 1], c = d[4 * i + 2], e = d[4 * i + 3];
        int e0 = a + c, e1 = a - c, e2 = (b >> 1) - e, e3 = b + (e >> 1);
        d[4 * i] = e0 + e3; d[4 * i + 1] = e1 + e2; d[4 * i + 2] = e1 - e2; d[4 * i + 3] = e0 - e3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int a = d[j], b = d[4 + j], c = d[8 + j], e = d[12 + j];
        int g0 = a + c, g1 = a - c, g2 = (b >> 1) - e, g3 = b + (e >> 1);
        d[j] = (g0 + g3 + 32) >> 6; d[4 + j] = (g1 + g2 + 32) >> 6; d[8 + j] = (g1 - g2 + 32) >> 6; d[12 + j] = (g0 - g3 + 32) >> 6;
    }
}
// blkIdx (coding order) <-> raster position of a luma 4x4 block inside the macroblock
__device__ __forceinline__ int blk_to_raster(int blk) { return ((((blk >> 1) & 1) + 2 * (blk >> 3)) << 2) | ((blk & 1) + 2 * ((blk >> 2) & 1)); }
__device__ __forceinline__ int raster_to_blk(int r) { int bx = r & 3, by = r >> 2; return (by >> 1) * 8 + (bx >> 1) * 4 + (by & 1) * 2 + (bx & 1); }

// loads / stores through address-space-1 pointers (global_* instructions, vmcnt only; a generic pointer gives FLAT instructions that also count on lgkmcnt)
typedef uint32_t jm_g2u __attribute__((ext_vector_type(2)));
typedef uint32_t jm_g4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 gld_u2(const void *p) { const jm_g2u v = *(const __attribute__((address_space(1))) jm_g2u *)p; return make_uint2(v.x, v.y); }
__device__ __forceinline__ void gst_u4(void *p, uint4 v) { const jm_g4u t = {v.x, v.y, v.z, v.w}; *(__attribute__((address_space(1))) jm_g4u *)p = t; }

// A MbRec read at a wave-uniform address: scalar loads (constant address space), the record stays in scalar registers and every condition on it is a
// scalar branch.  The job list is written by the host before the launch, never by a kernel.
struct MbWords { uint32_t w[8]; };       // w[3] = ref[0..3], w[4..7] = mv[0..3] (x | y << 16)
__device__ __forceinline__ MbWords load_mbrec_uniform(const MbRec *p) {
    typedef uint32_t v8u __attribute__((ext_vector_type(8)));
    const v8u v = *(const __attribute__((address_space(4))) v8u *)(uintptr_t)p;
    return __builtin_bit_cast(MbWords, v);
}

// A MbRec as eight dwords in registers.  All dynamically indexed fields are extracted with shifts / selects so the
// record never has to live in scratch memory.
struct MbW { uint32_t w[8]; };
__device__ __forceinline__ MbW load_mbw(const MbRec *p) {
    const uint4 *q = (const uint4 *)p;
    uint4 a = q[0], b = q[1];
    MbW m; m.w[0] = a.x; m.w[1] = a.y; m.w[2] = a.z; m.w[3] = a.w; m.w[4] = b.x; m.w[5] = b.y; m.w[6] = b.z; m.w[7] = b.w;
    return m;
}
__device__ __forceinline__ MbW select_mbw(bool c, const MbW &a, const MbW &b) {
    MbW m;
#pragma unroll
    for (int i = 0; i < 8; i++) m.w[i] = c ? a.w[i] : b.w[i];
    return m;
}
__device__ __forceinline__ int mbw_kind(const MbW &m) { return m.w[0] & 255; }
__device__ __forceinline__ int mbw_qp(const MbW &m) { return (m.w[0] >> 8) & 255; }
__device__ __forceinline__ int mbw_modes(const MbW &m) { return (m.w[0] >> 16) & 255; }
__device__ __forceinline__ int mbw_flags(const MbW &m) { return m.w[0] >> 24; }
__device__ __forceinline__ int mbw_cbp_blk(const MbW &m) { return m.w[1] & 0xffff; }
__device__ __forceinline__ int mbw_slice(const MbW &m) { return m.w[1] >> 24; }
__device__ __forceinline__ int mbw_ref(const MbW &m, int b8) { return (int)(int8_t)(m.w[3] >> (8 * b8)); }
__device__ __forceinline__ void mbw_mv(const PicParams &pp, const MbW &m, int rpos, int &mx, int &my) {
    if (mbw_flags(m) & MBF_MV_EXT) { const short *v = pp.mv_ext + ((size_t)m.w[4] + rpos) * 2; mx = v[0]; my = v[1]; }
    else {
        int b8 = (rpos >> 3) * 2 + ((rpos & 3) >> 1);
        uint32_t w = b8 == 0 ? m.w[4] : (b8 == 1 ? m.w[5] : (b8 == 2 ? m.w[6] : m.w[7]));
        mx = (int)(int16_t)(w & 0xffff); my = (int)(int16_t)(w >> 16);
    }
}
// reference surfaces (s0, s1; -1 = list unused) and vectors of 4x4 block rpos, for either motion representation
__device__ __forceinline__ void mbw_motion(const PicParams &pp, const MbW &m, int rpos, int &s0, int &s1, int &x0, int &y0, int &x1, int &y1) {
    int b8 = (rpos >> 3) * 2 + ((rpos & 3) >> 1);
    s0 = mbw_ref(m, b8); s1 = -1; x1 = y1 = 0;
    if (mbw_modes(m) & MBM_BIPRED) {
        const short *rec = pp.mv_ext + (size_t)m.w[4] * 2;
        x0 = rec[rpos * 2]; y0 = rec[rpos * 2 + 1]; x1 = rec[32 + rpos * 2]; y1 = rec[32 + rpos * 2 + 1];
        s1 = ((const int8_t *)(rec + 64))[b8];
    } else mbw_mv(pp, m, rpos, x0, y0);
}
// 8.7.2.1 boundary strength between 4x4 luma blocks rp (in macroblock p) and rq (in macroblock q), raster indices
// horizontal: the edge runs horizontally.  In a field picture (8.7.2.1) an intra macroblock edge gets 4 only when it is vertical, and the vertical vector
// difference that counts as one FRAME sample is 2 quarter field samples.
__device__ __forceinline__ int boundary_strength(const PicParams &pp, const MbW &p, int rp, const MbW &q, int rq, bool mb_edge, bool horizontal) {
    if (mbw_kind(p) != MB_INTER || mbw_kind(q) != MB_INTER) return mb_edge && !(pp.field && horizontal) ? 4 : 3;
    const int vlim = pp.field ? 2 : 4;
    if (((mbw_cbp_blk(p) >> raster_to_blk(rp)) & 1) || ((mbw_cbp_blk(q) >> raster_to_blk(rq)) & 1)) return 2;
    if (!((mbw_modes(p) | mbw_modes(q)) & MBM_BIPRED)) {       // one list on both sides (P slices)
        if (mbw_ref(p, (rp >> 3) * 2 + ((rp & 3) >> 1)) != mbw_ref(q, (rq >> 3) * 2 + ((rq & 3) >> 1))) return 1;
        int px, py, qx, qy; mbw_mv(pp, p, rp, px, py); mbw_mv(pp, q, rq, qx, qy);
        return (iabs(px - qx) >= 4 || iabs(py - qy) >= vlim) ? 1 : 0;
    }
    // B slices: compare the SETS of reference pictures and the vectors that go with them
    int p0, p1, q0, q1, px0, py0, px1, py1, qx0, qy0, qx1, qy1;
    mbw_motion(pp, p, rp, p0, p1, px0, py0, px1, py1); mbw_motion(pp, q, rq, q0, q1, qx0, qy0, qx1, qy1);
    int np = (p0 >= 0) + (p1 >= 0), nq = (q0 >= 0) + (q1 >= 0);
    if (np != nq) return 1;
    auto far = [vlim](int ax, int ay, int bx, int by) { return iabs(ax - bx) >= 4 || iabs(ay - by) >= vlim; };
    if (np == 1) {
        bool pl1 = p0 < 0, ql1 = q0 < 0;
        if ((pl1 ? p1 : p0) != (ql1 ? q1 : q0)) return 1;
        return far(pl1 ? px1 : px0, pl1 ? py1 : py0, ql1 ? qx1 : qx0, ql1 ? qy1 : qy0) ? 1 : 0;
    }
    if (!((p0 == q0 && p1 == q1) || (p0 == q1 && p1 == q0))) return 1;
    bool straight = far(px0, py0, qx0, qy0) || far(px1, py1, qx1, qy1), crossed = far(px0, py0, qx1, qy1) || far(px1, py1, qx0, qy0);
    if (p0 != p1) return (p0 == q0 ? straight : crossed) ? 1 : 0;
    return (straight && crossed) ? 1 : 0;
}

}  // namespace jmamd
