// jmcodec_amd/csrc/hevc_bs.h -- HEVC boundary strengths (ITU-T H.265 8.7.2.3 / 8.7.2.4) from the job lists, per 4x4 cell and per edge segment.
//
// Round 5 moved the derivation from the host (hevc_slice.cpp finish_picture: 9 % of the parse) to the device: k_hevc_bs_raster paints what the job lists say
// into maps of 4x4 cells, k_hevc_bs evaluates one 4-sample edge segment of the 8x8 grid per lane.  Round 6: the two pieces of logic live here as
// __host__ __device__ functions, so that tests/test_hevc_bs.py can run them on the CPU against a literal per-edge restatement of the clauses that works from the
// coding blocks themselves, not from maps (tests/native/hevc_bs_check.cpp).
//
//   maps: pu_map[cell] = index of the prediction-block piece (HevcPu) that covers the cell; four byte planes: [0] the cell lies in an intra block,
//   [1] in a luma transform block with cbf_luma = 1, [2] / [3] its left / top edge is an edge of such a block -- painted on the block's own first column / row
//   AND on the cells behind its last ones.  All plain stores of the value 1 / of a block's own index: no atomics needed.
//   an edge exists where a painted block ends or where two different prediction blocks meet; bS 2: p0 or q0 intra; 1: a transform edge with coefficients on
//   either side, or different motion (8.7.2.4); else 0.  Pieces of one prediction block compare equal motion and give 0; a transform edge between two blocks
//   without coefficients inside one prediction block gives 0 by the clause itself, so neither needs to be known.  What the lists cannot say is in
//   HevcCtb.db_flags.
// Part of the replacement for cuvidDecodePicture with codec_type 1 (/root/reference/nv_dec/nv_dec.cpp:33-41).
#pragma once
#include "hevc_jobs.h"
#include "mc_packed.h"        // JM_HD

namespace jmamd {
namespace hbs {

struct Maps { int w4, h4; uint32_t *pu_map; uint8_t *f_intra, *f_cbf, *f_l, *f_t; };
JM_HD Maps maps_of(int w, int h, uint32_t *pu_map, uint8_t *cell_flags) {
    Maps m; m.w4 = w >> 2; m.h4 = h >> 2; const int cells = m.w4 * m.h4;
    m.pu_map = pu_map; m.f_intra = cell_flags; m.f_cbf = cell_flags + cells; m.f_l = cell_flags + 2 * cells; m.f_t = cell_flags + 3 * cells;
    return m;
}
JM_HD void paint_pu(const Maps &m, int i, const HevcPu &pu) {
    const int cx0 = pu.x >> 2, cy0 = pu.y >> 2, nx = pu.w >> 2, ny = pu.h >> 2;
    for (int r = 0; r < ny; r++) for (int k = 0; k < nx; k++) m.pu_map[(cy0 + r) * m.w4 + cx0 + k] = (uint32_t)i;
}
// an n x n block at (x, y): its cells into `own` (f_intra or f_cbf), its four edges into the edge planes
JM_HD void mark(const Maps &m, int x, int y, int n, uint8_t *own) {
    const int cx0 = x >> 2, cy0 = y >> 2, nu = n >> 2, w4 = m.w4;
    for (int r = 0; r < nu; r++) {
        uint8_t *o = own + (cy0 + r) * w4 + cx0;
        for (int k = 0; k < nu; k++) o[k] = 1;
        m.f_l[(cy0 + r) * w4 + cx0] = 1;
        if (cx0 + nu < w4) m.f_l[(cy0 + r) * w4 + cx0 + nu] = 1;
    }
    for (int k = 0; k < nu; k++) { m.f_t[cy0 * w4 + cx0 + k] = 1; if (cy0 + nu < m.h4) m.f_t[(cy0 + nu) * w4 + cx0 + k] = 1; }
}
JM_HD int iabs_(int v) { return v < 0 ? -v : v; }
JM_HD bool far_apart(const int16_t *u, const int16_t *v) { return iabs_(u[0] - v[0]) >= 4 || iabs_(u[1] - v[1]) >= 4; }     // 8.7.2.4: >= 4 in quarter samples
// The strength of the edge segment whose first q sample is (x, y): dir 0 = a vertical edge (x a multiple of 8, four rows), dir 1 = a horizontal one.
// Bits 0-1: bS; bit 2 / 3: the samples of the p / q side are exempt from the loop filters (qp8 bit 7: pcm_loop_filter_disabled / transquant bypass).
template <class CtbPtr, class PuPtr, class QpPtr>
JM_HD int edge_strength(const Maps &m, int dir, int x, int y, int ctb_log2, int ctb_w, CtbPtr ctbs, PuPtr pus, QpPtr qp8, int w8) {
    int bs = 0;
    const int xp = dir ? x : x - 1, yp = dir ? y - 1 : y, w4 = m.w4;
    do {
        if (dir ? y == 0 : x == 0) break;                             // picture boundary
        const int lg = ctb_log2, cs = 1 << lg;
        const int fq = ctbs[(y >> lg) * ctb_w + (x >> lg)].db_flags, fp = ctbs[(yp >> lg) * ctb_w + (xp >> lg)].db_flags;
        if (fq & (HDB_DISABLED | HDB_CONCEALED)) break;
        if (((dir ? y : x) & (cs - 1)) == 0 && (fq & (dir ? HDB_NO_TOP : HDB_NO_LEFT))) break;
        const uint8_t *f_edge = dir ? m.f_t : m.f_l;
        const int q = (y >> 2) * w4 + (x >> 2), p = dir ? q - w4 : q - 1;
        const bool iq = m.f_intra[q] != 0, ip_cell = m.f_intra[p] != 0, ip = ip_cell || (fp & HDB_CONCEALED);
        const bool tu = f_edge[q] != 0;
        uint32_t ia = 0, ib = 0;
        if (!iq && !ip_cell) { ia = m.pu_map[q]; ib = m.pu_map[p]; }
        if (!tu && ia == ib) break;                                   // no transform edge, and the same prediction block (or one of them intra: tu is set then)
        if (iq || ip) { bs = 2; break; }
        if (tu && (m.f_cbf[q] || m.f_cbf[p])) { bs = 1; break; }
        if (ia == ib) break;
        const HevcPu a = pus[ia], b = pus[ib];
        const int na = (a.slot0 >= 0) + (a.slot1 >= 0), nb = (b.slot0 >= 0) + (b.slot1 >= 0);
        if (na != nb) { bs = 1; break; }
        if (na == 1) {
            const int ra = a.slot0 >= 0 ? a.slot0 : a.slot1, rb = b.slot0 >= 0 ? b.slot0 : b.slot1;
            const int16_t *va = a.slot0 >= 0 ? a.mv0 : a.mv1, *vb = b.slot0 >= 0 ? b.mv0 : b.mv1;
            bs = (ra != rb || far_apart(va, vb)) ? 1 : 0;
            break;
        }
        const bool straight = a.slot0 == b.slot0 && a.slot1 == b.slot1, crossed = a.slot0 == b.slot1 && a.slot1 == b.slot0;
        if (!straight && !crossed) { bs = 1; break; }
        const bool ds = far_apart(a.mv0, b.mv0) || far_apart(a.mv1, b.mv1), dc = far_apart(a.mv0, b.mv1) || far_apart(a.mv1, b.mv0);
        bs = (straight && crossed ? (ds && dc) : (straight ? ds : dc)) ? 1 : 0;
    } while (0);
    if (bs) {
        if (qp8[(yp >> 3) * w8 + (xp >> 3)] & 128) bs |= 4;
        if (qp8[(y >> 3) * w8 + (x >> 3)] & 128) bs |= 8;
    }
    return bs;
}

}  // namespace hbs
}  // namespace jmamd
