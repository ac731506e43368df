// jmcodec_amd/csrc/hevc_slice.cpp -- see hevc_slice.h.
#include "hevc_slice.h"
#include "hevc_tables.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace jmamd {

namespace {
inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
enum { PART_2Nx2N, PART_2NxN, PART_Nx2N, PART_NxN, PART_2NxnU, PART_2NxnD, PART_nLx2N, PART_nRx2N };

// scan tables (6.5.3 - 6.5.5): [scanIdx][log2 size 1..3][position] -> x | y << 4
struct ScanTables {
    uint8_t t[3][4][64];
    uint8_t inv[3][4][64];            // position x | y << 3 -> scan index
    uint8_t sigpat[3][5][16];         // sigCtx pattern term (9.3.4.2.5) by [scanIdx][prevCsbf 0..3, 4 = the 4x4 map][scan position in the sub-block]
    ScanTables() {
        for (int l = 0; l <= 3; l++) {
            int n = 1 << l, k = 0;
            for (int s = 0; s <= 2 * (n - 1); s++) for (int x = 0; x <= s; x++) { int y = s - x; if (x < n && y < n) t[0][l][k++] = (uint8_t)(x | (y << 4)); }
            for (int i = 0; i < n * n; i++) { t[1][l][i] = (uint8_t)((i & (n - 1)) | ((i >> l) << 4)); t[2][l][i] = (uint8_t)((i >> l) | ((i & (n - 1)) << 4));
                }
            for (int sidx = 0; sidx < 3; sidx++) for (int i = 0; i < n * n; i++) inv[sidx][l][(t[sidx][l][i] & 15) | ((t[sidx][l][i] >> 4) << 3)] = (uint8_t)i;
        }
        static const uint8_t map4[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8};
        for (int sidx = 0; sidx < 3; sidx++) for (int k = 0; k < 16; k++) {
            const int xq = t[sidx][2][k] & 15, yq = t[sidx][2][k] >> 4;
            sigpat[sidx][0][k] = (uint8_t)(xq + yq == 0 ? 2 : (xq + yq < 3 ? 1 : 0));
            sigpat[sidx][1][k] = (uint8_t)(yq == 0 ? 2 : (yq == 1 ? 1 : 0));
            sigpat[sidx][2][k] = (uint8_t)(xq == 0 ? 2 : (xq == 1 ? 1 : 0));
            sigpat[sidx][3][k] = 2;
            sigpat[sidx][4][k] = map4[(yq << 2) + xq];
        }
    }
};
const ScanTables kScan;
// A run of per-4x4 map entries of one value, n = 1 .. 16 units (a row of a coding / transform unit).  The library's memset costs more than the stores at
// these sizes (its masked-store path for n < 32 was 4 % of the parse): the power-of-two sizes, which is all a quadtree produces, are inline stores.
static inline void fill_units(void *p, int v, int n) {
    switch (n) {
    case 1: *(uint8_t *)p = (uint8_t)v; break;
    case 2: memset(p, v, 2); break;
    case 4: memset(p, v, 4); break;
    case 8: memset(p, v, 8); break;
    case 16: memset(p, v, 16); break;
    default: memset(p, v, (size_t)n); break;
    }
}
struct Flat16 { uint8_t v[32 * 32]; Flat16() { memset(v, 16, sizeof v); } };     // m[x][y] = 16 (7.3.4: scaling lists off, transform skip of a larger block)
const Flat16 kFlat16;

inline int mv_scale(int mv, int td, int tb) {                         // (8-179) ff.
    td = clip3(-128, 127, td); tb = clip3(-128, 127, tb);
    int tx = (16384 + (std::abs(td) >> 1)) / td, f = clip3(-4096, 4095, (tb * tx + 32) >> 6), p = f * mv;
    return clip3(-32768, 32767, p < 0 ? -((127 - p) >> 8) : (p + 127) >> 8);
}
inline bool same_motion(const HevcMotion &a, const HevcMotion &b) {
    if (a.pf != b.pf) return false;
    for (int l = 0; l < 2; l++) if ((a.pf >> l) & 1) if (a.ref[l] != b.ref[l] || a.mv[l][0] != b.mv[l][0] || a.mv[l][1] != b.mv[l][1]) return false;
    return true;
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------
void HevcPicParser::begin_picture(const HevcSps &sps, const HevcPps &pps, int poc, HevcPicJobs *jobs, HevcDigest *dgst, HevcColMotion *col_out) {
    sps_ = &sps; pps_ = &pps; poc_ = poc; jobs_ = jobs; dg_ = dgst; col_out_ = col_out; exported_rows_ = 0;
    w_ = sps.width; h_ = sps.height; w4_ = w_ >> 2; h4_ = h_ >> 2; ctb_size_ = 1 << sps.log2_ctb;
    ctb_w_ = (w_ + ctb_size_ - 1) >> sps.log2_ctb; ctb_h_ = (h_ + ctb_size_ - 1) >> sps.log2_ctb;
    const size_t n4 = (size_t)w4_ * h4_, nc = (size_t)ctb_w_ * ctb_h_;
    // The per-4x4 maps need no clearing between pictures: every read is guarded by the availability process (6.4.1), i.e. only touches
    // units this picture has already decoded, and coding_unit() rewrites every map for all units of the CU; finish_picture() fills in
    // coding tree blocks that no slice delivered.
    if (pm_.size() != n4) {
        pm_.assign(n4, 0); skip_.assign(n4, 0); depth_.assign(n4, 0); ipm_.assign(n4, 1); nofilter_.assign(n4, 0);
        qp_.assign(n4, 26); mot_.assign(n4, HevcMotion());
    }
    ctb_slice_.assign(nc, -1); ctb_sidx_.assign(nc, 0);
    slices_.clear(); wpp_valid_ = dep_valid_ = false; last_cu_qp_ = 26; err_ = false;
    // 6.5.1 raster <-> tile scan and 6.5.2 z-scan order: only when the layout changed
    uint64_t key = ((uint64_t)w_ << 48) ^ ((uint64_t)h_ << 32) ^ ((uint64_t)sps.log2_ctb << 28) ^ ((uint64_t)sps.log2_min_tb << 24) ^
        ((uint64_t)pps.tile_cols << 16) ^ ((uint64_t)pps.tile_rows << 8) ^ (pps.uniform ? 1 : 0);
    if (!pps.uniform) for (int i = 0; i < 20; i++) key = key * 1099511628211ULL + (uint64_t)(pps.col_w[i % 20] * 131 + pps.row_h[i % 22]);
    if (key != layout_key_ || rs2ts_.size() != nc) {
    layout_key_ = key;
    rs2ts_.resize(nc); ts2rs_.resize(nc); tile_id_.resize(nc);
    int colb[21], rowb[23];
    const int nc_t = pps.tile_cols, nr_t = pps.tile_rows;
    colb[0] = rowb[0] = 0;
    for (int i = 0; i < nc_t; i++) colb[i + 1] = pps.uniform ? ((i + 1) * ctb_w_) / nc_t : (i + 1 < nc_t ? std::min(ctb_w_, colb[i] + pps.col_w[i]) : ctb_w_);
    for (int i = 0; i < nr_t; i++) rowb[i + 1] = pps.uniform ? ((i + 1) * ctb_h_) / nr_t : (i + 1 < nr_t ? std::min(ctb_h_, rowb[i] + pps.row_h[i]) : ctb_h_);
    colb[nc_t] = ctb_w_; rowb[nr_t] = ctb_h_;
    int ts = 0;
    for (int tr = 0; tr < nr_t; tr++) for (int tc = 0; tc < nc_t; tc++)
        for (int y = rowb[tr]; y < rowb[tr + 1]; y++) for (int x = colb[tc]; x < colb[tc + 1]; x++) { int rs = y * ctb_w_ + x; rs2ts_[rs] = ts;
            ts2rs_[ts] = rs; tile_id_[ts] = tr * nc_t + tc; ts++; }
    layout_bad_ = ts != (int)nc;                                       // tile boundaries that do not cover the picture
    const int sh = sps.log2_ctb - sps.log2_min_tb;
    tb_w_ = ctb_w_ << sh;
    zs_.resize((size_t)tb_w_ * (ctb_h_ << sh));
    for (int y = 0; y < (ctb_h_ << sh); y++) for (int x = 0; x < tb_w_; x++) {
        uint32_t v = (uint32_t)rs2ts_[(y >> sh) * ctb_w_ + (x >> sh)] << (2 * sh);
        for (int i = 0; i < sh; i++) v |= (uint32_t)(((x >> i) & 1) << (2 * i)) | (uint32_t)(((y >> i) & 1) << (2 * i + 1));
        zs_[(size_t)y * tb_w_ + x] = v;
    }
    }
    if (layout_bad_) err_ = true;
    jobs->clear();
    jobs->ctbs.assign(nc, HevcCtb());
    if (col_out_) {
        HevcColMotion &c = *col_out_;
        c.w16 = (w_ + 15) >> 4; c.h16 = (h_ + 15) >> 4; c.poc = poc_;
        const size_t n = (size_t)c.w16 * c.h16;
        c.mot.resize(n); c.ref_poc.assign(2 * n, 0); c.lt.assign(n, 0); c.intra.assign(n, 1);
    }
}

void HevcPicParser::init_contexts() {                                  // 9.3.2.2
    const int t = sh_->type == HSL_I ? 0 : (sh_->type == HSL_P ? (sh_->cabac_init ? 2 : 1) : (sh_->cabac_init ? 1 : 2));
    const int qp = clip3(0, 51, sh_->qp);
    for (int i = 0; i < HEVC_N_CTX; i++) {
        int v = hevc_ctx_init[t][i], pre = clip3(1, 126, ((((v >> 4) * 5 - 45) * qp) >> 4) + ((v & 15) << 3) - 16);
        cb_.state[i] = pre <= 63 ? (Cabac::State)((63 - pre) << 1) : (Cabac::State)(((pre - 64) << 1) | 1);
    }
}

// 6.4.1 z-scan order availability
// The MinTbAddrZs comparison without its table (two dependent loads from half a megabyte at 1080p: 4 % of the parse in cache misses).  Inside one
// coding tree block the z-scan order of two positions is the order of their bit-interleaved coordinates (6.5.2); another coding tree block is
// available exactly when it has been decoded as part of this slice -- ctb_slice_ is set when a block's decoding starts, so a block that follows in
// decoding order does not carry this slice's address yet -- and lies in the same tile.
static inline uint32_t zorder4(uint32_t x, uint32_t y) {       // x, y < 16 (a 64x64 block in 4x4 units)
    static const uint8_t sp[16] = {0x00, 0x01, 0x04, 0x05, 0x10, 0x11, 0x14, 0x15, 0x40, 0x41, 0x44, 0x45, 0x50, 0x51, 0x54, 0x55};
    return (uint32_t)sp[x] | ((uint32_t)sp[y] << 1);
}
bool HevcPicParser::avail_zs(int xc, int yc, int xn, int yn) const {
    if (xn < 0 || yn < 0 || xn >= w_ || yn >= h_) return false;
    const int lc = sps_->log2_ctb;
    if ((((xn ^ xc) | (yn ^ yc)) >> lc) == 0) {                            // same coding tree block: same slice, same tile -- decoded already?
        const uint32_t m = (1u << lc) - 1;
        return zorder4(((uint32_t)xn & m) >> 2, ((uint32_t)yn & m) >> 2) <= zorder4(((uint32_t)xc & m) >> 2, ((uint32_t)yc & m) >> 2);
    }
    const int cn = (yn >> lc) * ctb_w_ + (xn >> lc);
    return ctb_slice_[cn] == sh_->slice_addr && rs2ts_[cn] < ctb_ts_ && tile_id_[rs2ts_[cn]] == tile_id_[ctb_ts_];
}
// 6.4.2 prediction block availability
bool HevcPicParser::avail_pb(int xcb, int ycb, int ncb, int xp, int yp, int w, int h, int part, int xn, int yn) const {
    bool a;
    if (xn >= xcb && yn >= ycb && xn < xcb + ncb && yn < ycb + ncb) a = !(2 * w == ncb && 2 * h == ncb && part == 1 && yn >= ycb + h && xn < xcb + w);
    else a = avail_zs(xp, yp, xn, yn);
    return a && pm_[i4(xn, yn)] == 1;
}

// ------------------------------------------------------------------------------------------------------------
// 8.6.1 luma quantisation parameter
void HevcPicParser::derive_qp(int xcb, int ycb) {
    const int lq = sps_->log2_ctb - pps_->diff_cu_qp_delta_depth, xq = xcb & ~((1 << lq) - 1), yq = ycb & ~((1 << lq) - 1);
    const int prev = first_qg_ ? sh_->qp : qp_prev_, cm = ~(ctb_size_ - 1);
    int a = prev, b = prev;
    if (((xq - 1) & cm) == (xq & cm) && avail_zs(xcb, ycb, xq - 1, yq)) a = qp_[i4(xq - 1, yq)];
    if (((yq - 1) & cm) == (yq & cm) && avail_zs(xcb, ycb, xq, yq - 1)) b = qp_[i4(xq, yq - 1)];
    qp_y_ = (((a + b + 1) >> 1) + dqp_ + 52) % 52;
}

// ------------------------------------------------------------------------------------------------------------
// 8.5.3.2.8 / 8.5.3.2.9
bool HevcPicParser::temporal(int xp, int yp, int w, int h, int X, int ridx, int16_t mv[2]) {
    if (!sh_->temporal_mvp || !refs_->col) return false;
    const HevcColMotion &col = *refs_->col;
    for (int pass = 0; pass < 2; pass++) {
        int xc = pass ? xp + (w >> 1) : xp + w, yc = pass ? yp + (h >> 1) : yp + h;
        if (!pass && ((yp >> sps_->log2_ctb) != (yc >> sps_->log2_ctb) || xc >= w_ || yc >= h_)) continue;
        const int e = (yc >> 4) * col.w16 + (xc >> 4);
        if (col.intra[e]) continue;
        const HevcMotion &cm = col.mot[e];
        int l;
        if (!(cm.pf & 1)) l = 1; else if (!(cm.pf & 2)) l = 0;
        else {
            bool no_backward = true;
            for (int k = 0; k < 2; k++) for (int i = 0; i < sh_->n_ref[k]; i++) if (refs_->poc[k][i] > poc_) no_backward = false;
            l = no_backward ? X : (sh_->col_from_l0 ? 1 : 0);
        }
        const int lt = (col.lt[e] >> l) & 1;
        if (lt != refs_->is_lt[X][ridx]) continue;
        const int cd = col.poc - col.ref_poc[2 * e + l], bd = poc_ - refs_->poc[X][ridx];
        if (lt || cd == bd || cd == 0) { mv[0] = cm.mv[l][0]; mv[1] = cm.mv[l][1]; }
        else { mv[0] = (int16_t)mv_scale(cm.mv[l][0], cd, bd); mv[1] = (int16_t)mv_scale(cm.mv[l][1], cd, bd); }
        return true;
    }
    return false;
}

// 8.5.3.2.2 - 8.5.3.2.5: the list is only built as far as the wanted index
int HevcPicParser::merge_candidates(int xcb, int ycb, int ncb, int xp, int yp, int w, int h, int part, int want, HevcMotion *list) {
    const int pl = pps_->log2_par_mrg, max = sh_->max_merge;
    int pmode = part_mode_;
    if (pl > 2 && ncb == 8) { xp = xcb; yp = ycb; w = h = 8; part = 0; pmode = PART_2Nx2N; }
    const int nx[5] = {xp - 1, xp + w - 1, xp + w, xp - 1, xp - 1}, ny[5] = {yp + h - 1, yp - 1, yp - 1, yp + h, yp - 1};    // A1 B1 B0 A0 B2
    bool have[5]; HevcMotion c[5];
    for (int k = 0; k < 5; k++) {
        have[k] = !((xp >> pl) == (nx[k] >> pl) && (yp >> pl) == (ny[k] >> pl)) && avail_pb(xcb, ycb, ncb, xp, yp, w, h, part, nx[k], ny[k]);
        if (k == 0 && part == 1 && (pmode == PART_Nx2N || pmode == PART_nLx2N || pmode == PART_nRx2N)) have[k] = false;
        if (k == 1 && part == 1 && (pmode == PART_2NxN || pmode == PART_2NxnU || pmode == PART_2NxnD)) have[k] = false;
        if (have[k]) c[k] = mot_[i4(nx[k], ny[k])];
    }
    // 8.5.3.2.3: B0 and B2 are compared with B1 whenever B1 is AVAILABLE (availableB1), also when B1 itself was dropped as a duplicate of A1
    // (availableFlagB1 = 0); only the count of four uses the flags
    const bool avail_b1 = have[1];
    if (have[1] && have[0] && same_motion(c[1], c[0])) have[1] = false;
    if (have[2] && avail_b1 && same_motion(c[2], c[1])) have[2] = false;
    if (have[3] && have[0] && same_motion(c[3], c[0])) have[3] = false;
    if (have[4] && (have[0] + have[1] + have[2] + have[3] == 4 || (have[0] && same_motion(c[4], c[0])) || (avail_b1 && same_motion(c[4],
        c[1])))) have[4] = false;
    int n = 0;
    for (int k = 0; k < 5; k++) if (have[k]) list[n++] = c[k];
    if (n > want && n <= max) return n;
    if (n < max && sh_->temporal_mvp) {
        HevcMotion t; memset(&t, 0, sizeof t); t.ref[0] = t.ref[1] = -1;
        if (temporal(xp, yp, w, h, 0, 0, t.mv[0])) { t.pf |= 1; t.ref[0] = 0; }
        if (sh_->type == HSL_B && temporal(xp, yp, w, h, 1, 0, t.mv[1])) { t.pf |= 2; t.ref[1] = 0; }
        if (t.pf) list[n++] = t;
    }
    if (n > max) n = max;
    if (sh_->type == HSL_B && n > 1 && n < max) {
        static const uint8_t a0[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, a1[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
        const int orig = n;
        for (int k = 0; k < orig * (orig - 1) && n < max; k++) {
            const HevcMotion &p = list[a0[k]], &q = list[a1[k]];
            if (!(p.pf & 1) || !(q.pf & 2)) continue;
            if (refs_->poc[0][p.ref[0]] == refs_->poc[1][q.ref[1]] && p.mv[0][0] == q.mv[1][0] && p.mv[0][1] == q.mv[1][1]) continue;
            HevcMotion t; t.pf = 3; t.pad = 0; t.ref[0] = p.ref[0]; t.ref[1] = q.ref[1];
            t.mv[0][0] = p.mv[0][0]; t.mv[0][1] = p.mv[0][1]; t.mv[1][0] = q.mv[1][0]; t.mv[1][1] = q.mv[1][1];
            list[n++] = t;
        }
    }
    const int nr = sh_->type == HSL_P ? sh_->n_ref[0] : std::min(sh_->n_ref[0], sh_->n_ref[1]);
    for (int z = 0; n < max; z++) {
        HevcMotion t; memset(&t, 0, sizeof t);
        t.pf = sh_->type == HSL_P ? 1 : 3; t.ref[0] = (int8_t)(z < nr ? z : 0); t.ref[1] = (int8_t)(sh_->type == HSL_P ? -1 : (z < nr ? z : 0));
        list[n++] = t;
    }
    return n;
}

// 8.5.3.2.6 / 8.5.3.2.7
void HevcPicParser::amvp(int xcb, int ycb, int ncb, int xp, int yp, int w, int h, int part, int X, int ridx, int flag, int16_t out[2]) {
    const int tp = refs_->poc[X][ridx], tl = refs_->is_lt[X][ridx];
    const int ax[2] = {xp - 1, xp - 1}, ay[2] = {yp + h, yp + h - 1}, bx[3] = {xp + w, xp + w - 1, xp - 1}, by[3] = {yp - 1, yp - 1, yp - 1};
    bool oka[2], okb[3], got_a = false, got_b = false; int16_t A[2] = {0, 0}, B[2] = {0, 0};
    for (int k = 0; k < 2; k++) oka[k] = avail_pb(xcb, ycb, ncb, xp, yp, w, h, part, ax[k], ay[k]);
    for (int k = 0; k < 3; k++) okb[k] = avail_pb(xcb, ycb, ncb, xp, yp, w, h, part, bx[k], by[k]);
    // neighbours lie in the current slice (6.4.1), so their reference indices refer to this slice's lists
    auto same_ref = [&](const HevcMotion &m, int16_t *mv) {
        for (int t = 0; t < 2; t++) { int l = t ? !X : X; if (((m.pf >> l) & 1) && refs_->poc[l][m.ref[l]] == tp) { mv[0] = m.mv[l][0]; mv[1] = m.mv[l][1];
            return true; } }
        return false;
    };
    auto scaled_ref = [&](const HevcMotion &m, int16_t *mv) {
        for (int t = 0; t < 2; t++) {
            int l = t ? !X : X;
            if (!((m.pf >> l) & 1) || refs_->is_lt[l][m.ref[l]] != tl) continue;
            mv[0] = m.mv[l][0]; mv[1] = m.mv[l][1];
            const int td = poc_ - refs_->poc[l][m.ref[l]], tb = poc_ - tp;
            if (!tl && td != tb && td != 0) { mv[0] = (int16_t)mv_scale(mv[0], td, tb); mv[1] = (int16_t)mv_scale(mv[1], td, tb); }
            return true;
        }
        return false;
    };
    for (int k = 0; k < 2 && !got_a; k++) if (oka[k]) got_a = same_ref(mot_[i4(ax[k], ay[k])], A);
    for (int k = 0; k < 2 && !got_a; k++) if (oka[k]) got_a = scaled_ref(mot_[i4(ax[k], ay[k])], A);
    for (int k = 0; k < 3 && !got_b; k++) if (okb[k]) got_b = same_ref(mot_[i4(bx[k], by[k])], B);
    if (!oka[0] && !oka[1]) {
        if (got_b) { got_a = true; A[0] = B[0]; A[1] = B[1]; }
        got_b = false;
        for (int k = 0; k < 3 && !got_b; k++) if (okb[k]) got_b = scaled_ref(mot_[i4(bx[k], by[k])], B);
    }
    int16_t list[3][2]; int n = 0;
    if (got_a) { list[n][0] = A[0]; list[n][1] = A[1]; n++; }
    if (got_b && !(got_a && A[0] == B[0] && A[1] == B[1])) { list[n][0] = B[0]; list[n][1] = B[1]; n++; }
    if (n < 2 && n <= flag) { int16_t t[2]; if (temporal(xp, yp, w, h, X, ridx, t)) { list[n][0] = t[0]; list[n][1] = t[1]; n++; } }
    for (; n < 2; n++) list[n][0] = list[n][1] = 0;
    out[0] = list[flag][0]; out[1] = list[flag][1];
}

// ------------------------------------------------------------------------------------------------------------
// 7.3.8.3
void HevcPicParser::parse_sao(int rs) {
    HevcCtb &o = jobs_->ctbs[rs];
    if (!sh_->sao_luma && !sh_->sao_chroma) return;
    const int rx = rs % ctb_w_, ry = rs / ctb_w_;
    bool left = false, up = false;
    if (rx > 0 && ctb_slice_[rs - 1] == sh_->slice_addr && tile_id_[rs2ts_[rs - 1]] == tile_id_[ctb_ts_]) left = cb_.decision(HEVC_CTX_SAO_MERGE);
    if (!left && ry > 0 && ctb_slice_[rs - ctb_w_] == sh_->slice_addr &&
        tile_id_[rs2ts_[rs - ctb_w_]] == tile_id_[ctb_ts_]) up = cb_.decision(HEVC_CTX_SAO_MERGE);
    if (left || up) {
        const HevcCtb &s = jobs_->ctbs[left ? rs - 1 : rs - ctb_w_];
        memcpy(o.sao_type, s.sao_type, 3); memcpy(o.sao_pos, s.sao_pos, 3); memcpy(o.sao_off, s.sao_off, 12);
    } else for (int c = 0; c < 3; c++) {
        if (!(c ? sh_->sao_chroma : sh_->sao_luma)) continue;
        if (c == 2) o.sao_type[2] = o.sao_type[1];
        else o.sao_type[c] = cb_.decision(HEVC_CTX_SAO_TYPE) ? (cb_.bypass() ? 2 : 1) : 0;
        if (!o.sao_type[c]) continue;
        int a[4];
        for (int i = 0; i < 4; i++) { a[i] = 0; while (a[i] < 7 && cb_.bypass()) a[i]++; }
        if (o.sao_type[c] == 1) {
            for (int i = 0; i < 4; i++) if (a[i] && cb_.bypass()) a[i] = -a[i];
            int p = 0; for (int i = 0; i < 5; i++) p = (p << 1) | cb_.bypass();
            o.sao_pos[c] = (uint8_t)p;
        } else {
            a[2] = -a[2]; a[3] = -a[3];
            if (c == 2) o.sao_pos[2] = o.sao_pos[1]; else { int p = cb_.bypass(); p = (p << 1) | cb_.bypass(); o.sao_pos[c] = (uint8_t)p; }
        }
        for (int i = 0; i < 4; i++) o.sao_off[c][i] = (int8_t)a[i];
    }
    if (!sh_->sao_luma) o.sao_type[0] = 0;
    if (!sh_->sao_chroma) o.sao_type[1] = o.sao_type[2] = 0;
    if (o.sao_type[0] | o.sao_type[1] | o.sao_type[2]) jobs_->any_sao = true;
    if (dg_->on) for (int c = 0; c < 3; c++) { dg(0x6000 | (c << 8) | (o.sao_type[c] << 6) | o.sao_pos[c] * (o.sao_type[c] != 0));
        if (o.sao_type[c]) for (int i = 0; i < 4; i++) dg(o.sao_off[c][i]); }
}

// intra transform block record; xp/yp in samples of plane c
// 6.4.1 for the neighbours of an intra block, all at once.  Inside the block's own coding tree block the answer depends on nothing but the positions (z-scan
// order): a table by (block size, position in the CTB) built once per CTB size holds, for the 4-sample units down the left edge and along the top edge, who
// precedes the block.  Units in a neighbouring CTB are available when that CTB is (same slice, same tile, decoded: four flags per CTB, looked up once per CTB),
// units below the CTB / to its right never are, units outside the picture are cut off by count.  (One avail_zs() call per unit -- up to 33 per block -- was
// ~5 % of the parse.)
namespace {
struct IntraNb { uint16_t left, top; };                                // bit i: unit i down the left edge / along the top edge precedes the block, same CTB
struct IntraNbTable {
    IntraNb t[3][4][16 * 16];                                          // [log2_ctb - 4][log2 of the block's size in units][uy * U + ux]
    IntraNbTable() {
        for (int lc = 4; lc <= 6; lc++) { const int U = 1 << (lc - 2);
            for (int si = 0; si < 4; si++) { const int units = 2 << si;
                for (int uy = 0; uy < U; uy++) for (int ux = 0; ux < U; ux++) {
                    IntraNb e = {0, 0}; const uint32_t cur = zorder4((uint32_t)ux, (uint32_t)uy);
                    for (int i = 0; i < units && i < 16; i++) {
                        if (ux > 0 && uy + i < U && zorder4((uint32_t)(ux - 1), (uint32_t)(uy + i)) <= cur) e.left |= (uint16_t)(1u << i);
                        if (uy > 0 && ux + i < U && zorder4((uint32_t)(ux + i), (uint32_t)(uy - 1)) <= cur) e.top |= (uint16_t)(1u << i);
                    }
                    t[lc - 4][si][uy * U + ux] = e;
                } } }
    }
};
const IntraNbTable kIntraNb;
inline uint32_t low_bits(int n) { return n >= 32 ? 0xffffffffu : (1u << (n < 0 ? 0 : n)) - 1u; }
}  // namespace

// which of the four neighbouring coding tree blocks the current one may predict from (6.4.1: decoded, same slice, same tile): bit 0 left, 1 above, 2 above
// right, 3 above left
int HevcPicParser::ctb_neighbours() {
    if (nbf_rs_ == ctb_rs_) return nbf_;
    const int cx = ctb_rs_ % ctb_w_, cy = ctb_rs_ / ctb_w_;
    auto ok = [&](int cn) { return ctb_slice_[cn] == sh_->slice_addr && rs2ts_[cn] < ctb_ts_ && tile_id_[rs2ts_[cn]] == tile_id_[ctb_ts_]; };
    int f = 0;
    if (cx > 0 && ok(ctb_rs_ - 1)) f |= 1;
    if (cy > 0 && ok(ctb_rs_ - ctb_w_)) f |= 2;
    if (cy > 0 && cx + 1 < ctb_w_ && ok(ctb_rs_ - ctb_w_ + 1)) f |= 4;
    if (cy > 0 && cx > 0 && ok(ctb_rs_ - ctb_w_ - 1)) f |= 8;
    nbf_rs_ = ctb_rs_; nbf_ = f;
    return f;
}

void HevcPicParser::emit_intra_tb(int xp, int yp, int log2, int c, int mode, bool with_coefs) {
    HevcIntraTb t; memset(&t, 0, sizeof t);
    const int n = 1 << log2, sc = c ? 1 : 0, xl = xp << sc, yl = yp << sc, unit = 4;
    t.x = (uint16_t)xp; t.y = (uint16_t)yp; t.log2 = (uint8_t)log2; t.plane = (uint8_t)c; t.mode = (uint8_t)mode;
    if (mode != kHevcModePcm) {
        const bool cip = pps_->constrained_intra;
        const int units = (2 * n << sc) / unit;                       // 4-luma-sample units along each of the two edges
        const int lc = sps_->log2_ctb, si = log2 + sc - 2;
        if (!cip && lc >= 4 && lc <= 6 && si >= 0 && si <= 3) {
            const int U = 1 << (lc - 2), cm = (1 << lc) - 1, ux = (xl & cm) >> 2, uy = (yl & cm) >> 2, nb = ctb_neighbours();
            const IntraNb e = kIntraNb.t[lc - 4][si][uy * U + ux];
            uint32_t left = e.left, top = e.top;
            if (ux == 0 && (nb & 1)) left |= low_bits(std::min(units, U - uy));                   // the CTB to the left (the one below it is never decoded yet)
            if (uy == 0) {
                const int inside = std::min(units, U - ux);
                if (nb & 2) top |= low_bits(inside);                                               // the CTB above
                if ((nb & 4) && units > inside) top |= low_bits(units) & ~low_bits(inside);        // ... and the one above right
            }
            left &= low_bits(std::min(units, (h_ - yl + 3) >> 2));                                  // units below / right of the picture
            top &= low_bits(std::min(units, (w_ - xl + 3) >> 2));
            t.avail = left | top << 16;
            if ((ux > 0 && uy > 0) || (ux == 0 && uy == 0 ? (nb & 8) : ux == 0 ? (nb & 1) : (nb & 2))) t.flags |= HTB_CORNER;
#ifdef JM_CHECK_INTRA_NB                                               // developer cross-check against the unit-by-unit form (tools/host_bench with HB_FLAGS)
            { uint32_t av = 0; bool corner = avail_zs(xl, yl, xl - 1, yl - 1);
              for (int i = 0; i < units; i++) { if (avail_zs(xl, yl, xl - 1, yl + i * unit)) av |= 1u << i;
                  if (avail_zs(xl, yl, xl + i * unit, yl - 1)) av |= 1u << (16 + i); }
              if (av != t.avail || corner != ((t.flags & HTB_CORNER) != 0)) {
                  fprintf(stderr, "intra neighbours differ at (%d, %d) log2 %d plane %d: %08x / %08x\n", xl, yl, log2, c, t.avail, av); abort(); } }
#endif
        } else {
            for (int i = 0; i < units; i++) {
                int yy = yl + i * unit, xx = xl + i * unit;
                if (avail_zs(xl, yl, xl - 1, yy) && (!cip || pm_[i4(xl - 1, yy)] == 2)) t.avail |= 1u << i;
                if (avail_zs(xl, yl, xx, yl - 1) && (!cip || pm_[i4(xx, yl - 1)] == 2)) t.avail |= 1u << (16 + i);
            }
            if (avail_zs(xl, yl, xl - 1, yl - 1) && (!cip || pm_[i4(xl - 1, yl - 1)] == 2)) t.flags |= HTB_CORNER;
        }
    }
    t.coef_off = (uint32_t)jobs_->coefs.size();
    (void)with_coefs;
    jobs_->itbs.push_back(t);
}

// ------------------------------------------------------------------------------------------------------------
// The entropy-coded part of one 4x4 sub-block (7.3.8.11 from sig_coeff_flag to coeff_abs_level_remaining; 9.3.3.11, 9.3.4.2.5 - 9.3.4.2.7) as a function of
// its own: its loops carry the arithmetic decoder's variables from bin to bin, and inside residual_coding -- next to the block's positions, scaling and
// output variables -- the compiler kept them on the stack (55 instructions per significance bin).  Nothing here branches on a decoded bin except the loop
// ends: the greater-than-1 flags of the first eight coefficients are collected in a bit mask, which also says who carries a remaining level.
namespace {
struct SubBlock {
    // in
    uint32_t sig;                  // the last significant position of the block, when it lies in this sub-block
    int start;                     // first scan position to decode a flag for (downwards; -1: none)
    bool infer_dc;                 // the sub-block is coded and not the first or last one: position 0 is significant when no other position is
    const uint8_t *pat;            // context increment of sig_coeff_flag by scan position
    int sig_ctx, dc_ctx, cset_base;
    bool chroma, first_group, sign_hiding;
    // in / out
    int g1ctx;                     // greater1Ctx at the end of the previous coded sub-block
    // out
    int np;                        // significant coefficients; m = 0 .. np - 1 counts them in decoding order (highest scan position first)
    uint32_t g1, g2, signs;        // bit m: coefficient m is greater than 1 / than 2; signs: coefficient 0 in bit 31, a hidden sign reads as 0
    bool hide;
    int remv[16];                  // coeff_abs_level_remaining by m (0 where none is coded)
};
__attribute__((noinline)) bool decode_sub_block(Cabac &home, SubBlock &sb) {
    CabacRegs cb(home);
    uint32_t sig = sb.sig;
    {
        const uint8_t *pat = sb.pat; const int ctx = sb.sig_ctx;
        for (int k = sb.start; k >= 1; k--) sig |= (uint32_t)cb.decision(ctx + pat[k]) << k;
        if (sb.start >= 0) {
            if (sb.infer_dc && !sig) sig = 1;
            else sig |= (uint32_t)cb.decision(sb.dc_ctx);
        }
    }
    sb.sig = sig;
    if (!sig) { cb.commit(); return true; }
    const int np = __builtin_popcount(sig), n8 = np < 8 ? np : 8;
    int cset = sb.cset_base;
    if (!sb.first_group && sb.g1ctx == 0) cset++;
    uint32_t g1 = 0, g1ctx = 1;
    const int g1base = HEVC_CTX_G1 + cset * 4 + (sb.chroma ? 16 : 0);
    for (int m = 0; m < n8; m++) {
        const uint32_t b = (uint32_t)cb.decision(g1base + (int)g1ctx);
        g1 |= b << m;
        g1ctx = (g1ctx + ((g1ctx - 1) < 2u)) & (b - 1);             // 1 -> 2 -> 3 while the flags are 0, 0 from the first 1 on
    }
    uint32_t g2 = 0;                                                // the greater-than-2 flag belongs to the first coefficient greater than 1
    if (g1 && cb.decision(HEVC_CTX_G2 + cset + (sb.chroma ? 4 : 0))) g2 = g1 & (0u - g1);
    // a remaining level follows when the flags could not say more: base level 3 for the coefficient with the greater-than-2 flag, 2 for the others of
    // the first eight, 1 from the ninth on
    uint32_t rem_mask = (g1 & (g1 - 1)) | g2 | (((1u << np) - 1) & ~0xffu);
    const int hi_pos = 31 - __builtin_clz(sig), lo_pos = __builtin_ctz(sig);
    const bool hide = sb.sign_hiding && hi_pos - lo_pos > 3;
    const int nsign = np - (hide ? 1 : 0);
    sb.signs = nsign ? cb.bypass_bits(nsign) << (32 - nsign) : 0;
    memset(sb.remv, 0, sizeof sb.remv);
    for (int rice = 0; rem_mask;) {
        const int m = __builtin_ctz(rem_mask); rem_mask &= rem_mask - 1;
        const int q = cb.unary(32);
        if (q >= 32) return false;
        int rem;
        if (q < 4) { rem = q << rice; if (rice) rem |= (int)cb.bypass_bits(rice); }
        else { const int nb = q - 3 + rice; if (nb > 30) return false;
            const int s = nb > 16 ? (int)(cb.bypass_bits(nb - 16) << 16 | cb.bypass_bits(16)) : (int)cb.bypass_bits(nb);
            rem = (((1 << (q - 3)) + 2) << rice) + s; }
        sb.remv[m] = rem;
        const int a = 1 + (int)((g1 >> m) & 1) + (int)((g2 >> m) & 1) + rem;
        if (a > 3 * (1 << rice)) rice = rice < 4 ? rice + 1 : 4;
    }
    sb.np = np; sb.g1 = g1; sb.g2 = g2; sb.hide = hide; sb.g1ctx = (int)g1ctx;
    cb.commit();                                                    // (the error returns above abandon the slice: nothing to write back)
    return true;
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------
// 7.3.8.11 residual_coding; (x0, y0) luma position for the digest, (xp, yp) position in plane c.  Appends the scaled coefficients.
bool HevcPicParser::residual_coding(int x0, int y0, int log2, int c, int xp, int yp, bool intra_tb) {
    const int n = 1 << log2;
    int tskip = 0;
    // last significant coefficient position
    int last[2];
    {
    CabacRegs cb(cb_);                                              // the arithmetic decoder's variables in registers (h264_cabac.h)
    if (pps_->transform_skip && !tq_bypass_ && log2 == 2) tskip = cb.decision(HEVC_CTX_TSKIP + (c ? 1 : 0));
    for (int d = 0; d < 2; d++) {
        const int cmax = 2 * log2 - 1, off = c ? 15 : 3 * (log2 - 2) + ((log2 - 1) >> 2), shf = c ? log2 - 2 : (log2 + 1) >> 2,
            base = d ? HEVC_CTX_LAST_Y : HEVC_CTX_LAST_X;
        int v = 0;
        while (v < cmax && cb.decision(base + off + (v >> shf))) v++;
        last[d] = v;
    }
    for (int d = 0; d < 2; d++) if (last[d] > 3) { int nb = (last[d] >> 1) - 1, s = 0; for (int i = 0; i < nb; i++) s = (s << 1) | cb.bypass();
        last[d] = (1 << nb) * (2 + (last[d] & 1)) + s; }
    cb.commit();
    }
    int scan = 0;
    if (cu_intra_ && (log2 == 2 || (log2 == 3 && c == 0))) { int pm = c == 0 ? ipm_[i4(x0, y0)] : ipm_c_; if (pm >= 6 && pm <= 14) scan = 2;
        else if (pm >= 22 && pm <= 30) scan = 1; }
    int lx = last[0], ly = last[1];
    if (scan == 2) std::swap(lx, ly);
    if (lx >= n || ly >= n) return false;
    const int nsl = log2 - 2, nsb = 1 << nsl;
    const uint8_t *sb_scan = kScan.t[scan][nsl], *pos_scan = kScan.t[scan][2];
    // locate the last position in scan order
    int last_sb = -1, last_pos = -1;
    last_sb = kScan.inv[scan][nsl][(lx >> 2) | ((ly >> 2) << 3)]; last_pos = kScan.inv[scan][2][(lx & 3) | ((ly & 3) << 3)];
    uint8_t csbf[8][8]; memset(csbf, 0, sizeof csbf);
    int g1ctx = 1; bool first_group = true;
    // 8.6.4.1 scaling.  Without the digest every level is scaled and appended to the coefficient list as it is decoded (`direct`); with it the levels are
    // first collected in lev_ / nz_pos_ (the digest wants them sorted by position) and scaled afterwards.
    const int qp = c == 0 ? qp_y_ : hevc_qpc_tab[clip3(0, 57, qp_y_ + (c == 1 ? pps_->cb_qp_off + sh_->cb_qp_off : pps_->cr_qp_off + sh_->cr_qp_off))];
    const HevcScaling &sf = pps_->scaling_present ? pps_->sf : sps_->sf;
    const int mat = (cu_intra_ ? 0 : 3) + c;
    const uint8_t *smat = log2 == 2 ? sf.f4[mat] : log2 == 3 ? sf.f8[mat] : log2 == 4 ? sf.f16[mat] : sf.f32[cu_intra_ ? 0 : 1];
    const bool flat = !sps_->scaling_enabled || (tskip && n > 4);
    const uint8_t *smul = flat ? kFlat16.v : smat;
    const int bd_shift = log2 + 3, ls = hevc_level_scale[qp % 6] << (qp / 6);
    const int64_t sc_add = (int64_t)1 << (bd_shift - 1);
    const bool direct = !dg_->on;
    uint32_t *const dw = jobs_->coefs.tail((size_t)n * n); uint32_t dcount = 0;      // the block's coefficients are written where they stay
    for (int k = 0; k < nz_n_; k++) lev_[nz_pos_[k]] = 0;             // lev_ is all zero between calls
    nz_n_ = 0;
    for (int i = last_sb; i >= 0; i--) {
        const int xs = sb_scan[i] & 15, ys = sb_scan[i] >> 4;
        const int right = xs < nsb - 1 ? csbf[ys][xs + 1] : 0, below = ys < nsb - 1 ? csbf[ys + 1][xs] : 0;
        bool infer_dc = false;
        if (i < last_sb && i > 0) { csbf[ys][xs] = (uint8_t)cb_.decision(HEVC_CTX_CSBF + ((right | below) ? 1 : 0) + (c ? 2 : 0)); infer_dc = true; }
        else csbf[ys][xs] = 1;
        if (!csbf[ys][xs]) continue;
        uint16_t sig = 0;                                              // bit k: position k of the sub-block is significant
        int start = 15;
        if (i == last_sb) { start = last_pos - 1; sig = (uint16_t)(1u << last_pos); }
        const int prev = right | (below << 1);
        // sig_coeff_flag context = per-sub-block base + a table term by scan position; only the DC of the whole block is special
        const uint8_t *pat = kScan.sigpat[scan][log2 == 2 ? 4 : prev];
        const int sig0 = HEVC_CTX_SIG + (c ? 27 : 0);
        const int sbase = log2 == 2 ? sig0 :
            (c == 0 ? HEVC_CTX_SIG + (i > 0 ? 3 : 0) + (log2 == 3 ? (scan == 0 ? 9 : 15) : 21) : HEVC_CTX_SIG + 27 + (log2 == 3 ? 9 : 12));
        SubBlock sb;
        sb.sig = sig; sb.start = start; sb.infer_dc = infer_dc; sb.pat = pat; sb.sig_ctx = sbase; sb.dc_ctx = (log2 > 2 && i == 0) ? sig0 : sbase + pat[0];
        sb.cset_base = ((i == 0 || c) ? 0 : 2); sb.chroma = c != 0; sb.g1ctx = g1ctx; sb.first_group = first_group;
        sb.sign_hiding = pps_->sign_hiding && !tq_bypass_;
        if (!decode_sub_block(cb_, sb)) return false;
        sig = (uint16_t)sb.sig;
        if (!sig) continue;
        first_group = false; g1ctx = sb.g1ctx;
        const int np = sb.np; const uint32_t g1 = sb.g1, g2 = sb.g2; uint32_t signs = sb.signs; const bool hide = sb.hide; const int *remv = sb.remv;
        const int sb_base = ((ys << 2) << log2) + (xs << 2), hidden = hide ? np - 1 : 99;
        int sum = 0;
        uint32_t left = sig;
        for (int m = 0; m < np; m++) {
            const int k = 31 - __builtin_clz(left); left ^= 1u << k;
            const int a = 1 + (int)((g1 >> m) & 1) + (int)((g2 >> m) & 1) + remv[m];
            sum += a;
            const int neg = (int)(signs >> 31) | ((m == hidden) & sum & 1); signs <<= 1;       // 9.3.4.3.6: the hidden sign is the parity of the sum
            const int idx = sb_base + ((pos_scan[k] >> 4) << log2) + (pos_scan[k] & 15);
            const int lv = clip3(-32768, 32767, (a ^ -neg) + neg);
            if (direct) {
                const int v = tq_bypass_ ? lv : clip3(-32768, 32767, (int)(((int64_t)lv * smul[idx] * ls + sc_add) >> bd_shift));
                dw[dcount] = (uint32_t)idx | ((uint32_t)(uint16_t)(int16_t)v << 16); dcount += v != 0;
            } else { lev_[idx] = (int16_t)lv; nz_pos_[nz_n_++] = (uint16_t)idx; }
        }
    }
    if (cb_.overrun) return false;
    if (dg_->on) {
        dg(0x7000 | (c << 8) | (log2 << 4) | tskip); dg(x0); dg(y0);
        std::sort(nz_pos_, nz_pos_ + nz_n_);
        for (int k = 0; k < nz_n_; k++) { dg(nz_pos_[k]); dg(lev_[nz_pos_[k]]); }
    }
    // 8.6.4.1 scaling -> sparse coefficient list (direct: done level by level above)
    const uint32_t first = (uint32_t)jobs_->coefs.size();
    uint32_t count = 0;
    if (direct) count = dcount;
    else {
        uint32_t *cw = dw;                                           // trimmed to what survived scaling
        if (tq_bypass_) {
            for (int k = 0; k < nz_n_; k++) { const int idx = nz_pos_[k], v = lev_[idx];
                if (v) cw[count++] = (uint32_t)idx | ((uint32_t)(uint16_t)(int16_t)v << 16); }
        } else {
            for (int k = 0; k < nz_n_; k++) {
                const int idx = nz_pos_[k];
                const int v = clip3(-32768, 32767, (int)(((int64_t)lev_[idx] * (flat ? 16 : smat[idx]) * ls + sc_add) >> bd_shift));
                if (v) cw[count++] = (uint32_t)idx | ((uint32_t)(uint16_t)(int16_t)v << 16);
            }
        }
    }
    jobs_->coefs.take(count);
    const uint8_t flags = (uint8_t)((tskip ? HTB_TSKIP : 0) | (tq_bypass_ ? HTB_BYPASS : 0) | ((cu_intra_ && c == 0 && n == 4) ? HTB_DST : 0));
    if (intra_tb) { HevcIntraTb &t = jobs_->itbs.back(); t.coef_off = first; t.coef_n = count; t.flags |= flags; }
    else if (count || c == 0) {                                     // (a luma block with cbf_luma = 1 is listed even when no level survived the scaling: bS, hevc_jobs.h)
        HevcTb t; t.x = (uint16_t)xp; t.y = (uint16_t)yp; t.log2 = (uint8_t)log2; t.plane = (uint8_t)c; t.flags = flags; t.pad = 0;
        t.coef_off = first; t.coef_n = count; jobs_->tbs.push_back(t); }
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// 7.3.8.8, 7.3.8.10
bool HevcPicParser::transform_unit(int x0, int y0, int xb, int yb, int log2, int depth, int blk, int cbf_y, int cbf_cb, int cbf_cr) {
    // (rounds 1-4 kept per-4x4 maps of transform / prediction edges and cbf_luma here for the boundary strengths: the device derives them from the job
    // lists now, hevc_kernels.hip k_hevc_bs_raster)
    if ((cbf_y || cbf_cb || cbf_cr) && pps_->cu_qp_delta && !dqp_coded_) {
        int v = 0;
        if (cb_.decision(HEVC_CTX_CU_QP_DELTA)) { v = 1; while (v < 5 && cb_.decision(HEVC_CTX_CU_QP_DELTA + 1)) v++; }
        if (v == 5) { int k = 0, a = 0; while (cb_.bypass()) { a += 1 << k; if (++k > 16) return false; }
            for (int b = k - 1; b >= 0; b--) a += cb_.bypass() << b; v += a; }
        if (v && cb_.bypass()) v = -v;
        if (v < -26 || v > 25) return false;
        dqp_coded_ = true; dqp_ = v;
        derive_qp(cu_x_, cu_y_);
    }
    if (cu_intra_) emit_intra_tb(x0, y0, log2, 0, ipm_[i4(x0, y0)], false);
    if (cbf_y && !residual_coding(x0, y0, log2, 0, x0, y0, cu_intra_)) return false;
    if (log2 > 2) {
        for (int c = 1; c < 3; c++) {
            if (cu_intra_) emit_intra_tb(x0 >> 1, y0 >> 1, log2 - 1, c, ipm_c_, false);
            if ((c == 1 ? cbf_cb : cbf_cr) && !residual_coding(x0, y0, log2 - 1, c, x0 >> 1, y0 >> 1, cu_intra_)) return false;
        }
    } else if (blk == 3) {
        for (int c = 1; c < 3; c++) {
            if (cu_intra_) emit_intra_tb(xb >> 1, yb >> 1, 2, c, ipm_c_, false);
            if ((c == 1 ? cbf_cb : cbf_cr) && !residual_coding(xb, yb, 2, c, xb >> 1, yb >> 1, cu_intra_)) return false;
        }
    }
    return true;
}
bool HevcPicParser::transform_tree(int x0, int y0, int xb, int yb, int log2, int depth, int blk, int pcb, int pcr) {
    bool split;
    if (log2 <= sps_->log2_max_tb && log2 > sps_->log2_min_tb && depth < max_tr_depth_ && !(intra_split_ &&
        depth == 0)) split = cb_.decision(HEVC_CTX_SPLIT_TF + 5 - log2);
    else split = log2 > sps_->log2_max_tb || (intra_split_ && depth == 0) || (sps_->depth_inter == 0 && !cu_intra_ && part_mode_ != PART_2Nx2N && depth == 0);
    int ccb = pcb, ccr = pcr;
    if (log2 > 2) { ccb = pcb ? cb_.decision(HEVC_CTX_CBF_CBCR + depth) : 0; ccr = pcr ? cb_.decision(HEVC_CTX_CBF_CBCR + depth) : 0; }
    if (split) {
        const int hh = 1 << (log2 - 1);
        for (int k = 0; k < 4; k++) if (!transform_tree(x0 + (k & 1) * hh, y0 + (k >> 1) * hh, x0, y0, log2 - 1, depth + 1, k, ccb, ccr)) return false;
        return true;
    }
    int cbf_y = 1;
    if (cu_intra_ || depth != 0 || ccb || ccr) cbf_y = cb_.decision(HEVC_CTX_CBF_LUMA + (depth == 0 ? 1 : 0));
    return transform_unit(x0, y0, xb, yb, log2, depth, blk, cbf_y, ccb, ccr);
}

// ------------------------------------------------------------------------------------------------------------
// 7.3.8.6, 7.3.8.9
bool HevcPicParser::prediction_unit(int xcb, int ycb, int ncb, int x0, int y0, int w, int h, int part_idx) {
    HevcMotion m; memset(&m, 0, sizeof m); m.ref[0] = m.ref[1] = -1;
    const bool merge = cu_skip_ ? true : (bool)cb_.decision(HEVC_CTX_MERGE_FLAG);
    last_merge_ = merge;
    if (merge) {
        int idx = 0;
        if (sh_->max_merge > 1 && cb_.decision(HEVC_CTX_MERGE_IDX)) { idx = 1; while (idx < sh_->max_merge - 1 && cb_.bypass()) idx++; }
        HevcMotion list[6];
        merge_candidates(xcb, ycb, ncb, x0, y0, w, h, part_idx, idx, list);
        m = list[idx];
        if (m.pf == 3 && w + h == 12) m.pf = 1;
        for (int l = 0; l < 2; l++) if (!((m.pf >> l) & 1)) { m.ref[l] = -1; m.mv[l][0] = m.mv[l][1] = 0; }
    } else {
        int idc = 0;
        if (sh_->type == HSL_B) {
            if (w + h != 12 && cb_.decision(HEVC_CTX_INTER_PRED_IDC + depth_[i4(x0, y0)])) idc = 2;
            else idc = cb_.decision(HEVC_CTX_INTER_PRED_IDC + 4);
        }
        int16_t mvd[2][2] = {{0, 0}, {0, 0}}; int flag[2] = {0, 0};
        for (int l = 0; l < 2; l++) {
            if (idc == (l ? 0 : 1)) continue;
            int ri = 0;
            if (sh_->n_ref[l] > 1) { const int cmax = sh_->n_ref[l] - 1;
                while (ri < cmax && (ri < 2 ? cb_.decision(HEVC_CTX_REF_IDX + ri) : cb_.bypass())) ri++; }
            m.ref[l] = (int8_t)ri; m.pf |= (uint8_t)(1 << l);
            if (!(l == 1 && sh_->mvd_l1_zero && idc == 2)) {
                int g0[2], g1[2] = {0, 0};
                g0[0] = cb_.decision(HEVC_CTX_MVD_G0); g0[1] = cb_.decision(HEVC_CTX_MVD_G0);
                if (g0[0]) g1[0] = cb_.decision(HEVC_CTX_MVD_G1);
                if (g0[1]) g1[1] = cb_.decision(HEVC_CTX_MVD_G1);
                for (int d = 0; d < 2; d++) {
                    int v = 0;
                    if (g0[d]) {
                        v = 1;
                        if (g1[d]) { int k = 1, a = 0; while (cb_.bypass()) { a += 1 << k; if (++k > 17) return false; }
                            for (int b = k - 1; b >= 0; b--) a += cb_.bypass() << b; v = a + 2; }
                        if (cb_.bypass()) v = -v;
                    }
                    if (v < -32768 || v > 32767) return false;
                    mvd[l][d] = (int16_t)v;
                }
            }
            flag[l] = cb_.decision(HEVC_CTX_MVP_FLAG);
        }
        for (int l = 0; l < 2; l++) if ((m.pf >> l) & 1) {
            int16_t p[2];
            amvp(xcb, ycb, ncb, x0, y0, w, h, part_idx, l, m.ref[l], flag[l], p);
            m.mv[l][0] = (int16_t)(p[0] + mvd[l][0]); m.mv[l][1] = (int16_t)(p[1] + mvd[l][1]);
        }
    }
    for (int l = 0; l < 2; l++) if (((m.pf >> l) & 1) && (m.ref[l] < 0 || m.ref[l] >= sh_->n_ref[l])) return false;
    {
        HevcMotion *mo = mot_.data();
        const int nux = w >> 2, nuy = h >> 2;
        for (int r = 0; r < nuy; r++) {
            const int i = i4(x0, y0 + 4 * r);
            for (int k = 0; k < nux; k++) mo[i + k] = m;
        }
    }
    if (dg_->on) { dg(0x5000 | (merge << 4) | m.pf); dg(x0); dg(y0); dg(w); dg(h); dg(m.ref[0]); dg(m.ref[1]); dg(m.mv[0][0]); dg(m.mv[0][1]); dg(m.mv[1][0]);
        dg(m.mv[1][1]); }
    // motion compensation jobs: tiles of at most 16x16 luma samples
    HevcPu j; memset(&j, 0, sizeof j);
    j.slot0 = (m.pf & 1) ? refs_->slot[0][m.ref[0]] : -1; j.slot1 = (m.pf & 2) ? refs_->slot[1][m.ref[1]] : -1;
    if (((m.pf & 1) && j.slot0 < 0) || ((m.pf & 2) && j.slot1 < 0)) return false;
    j.mv0[0] = m.mv[0][0]; j.mv0[1] = m.mv[0][1]; j.mv1[0] = m.mv[1][0]; j.mv1[1] = m.mv[1][1];
    j.ridx0 = (uint8_t)(m.ref[0] < 0 ? 0 : m.ref[0]); j.ridx1 = (uint8_t)(m.ref[1] < 0 ? 0 : m.ref[1]); j.wp = wp_index_;
    for (int y = 0; y < h; y += 16) for (int x = 0; x < w; x += 16) {
        j.x = (uint16_t)(x0 + x); j.y = (uint16_t)(y0 + y); j.w = (uint8_t)std::min(16, w - x); j.h = (uint8_t)std::min(16, h - y);
        jobs_->pus.push_back(j);
    }
    return true;
}

// 7.3.8.5
bool HevcPicParser::coding_unit(int x0, int y0, int log2) {
    const int n = 1 << log2;
    cu_x_ = x0; cu_y_ = y0; cu_intra_ = false; cu_skip_ = false; part_mode_ = PART_2Nx2N; tq_bypass_ = false; intra_split_ = false;
    dg_->n_cu++;
    if (pps_->tq_bypass) tq_bypass_ = cb_.decision(HEVC_CTX_CU_TQ_BYPASS);
    if (sh_->type != HSL_I) {
        const int inc = (avail_zs(x0, y0, x0 - 1, y0) && skip_[i4(x0 - 1, y0)]) + (avail_zs(x0, y0, x0, y0 - 1) && skip_[i4(x0, y0 - 1)]);
        cu_skip_ = cb_.decision(HEVC_CTX_CU_SKIP + inc);
    }
    if (!cu_skip_) {
        cu_intra_ = sh_->type == HSL_I ? true : (bool)cb_.decision(HEVC_CTX_PRED_MODE);
        if (!cu_intra_ || log2 == sps_->log2_min_cb) {
            if (cu_intra_) part_mode_ = cb_.decision(HEVC_CTX_PART_MODE) ? PART_2Nx2N : PART_NxN;
            else if (cb_.decision(HEVC_CTX_PART_MODE)) part_mode_ = PART_2Nx2N;
            else if (log2 == sps_->log2_min_cb) {
                if (cb_.decision(HEVC_CTX_PART_MODE + 1)) part_mode_ = PART_2NxN;
                else if (log2 == 3) part_mode_ = PART_Nx2N;
                else part_mode_ = cb_.decision(HEVC_CTX_PART_MODE + 2) ? PART_Nx2N : PART_NxN;
            } else if (!sps_->amp) part_mode_ = cb_.decision(HEVC_CTX_PART_MODE + 1) ? PART_2NxN : PART_Nx2N;
            else {
                const int hor = cb_.decision(HEVC_CTX_PART_MODE + 1);
                if (cb_.decision(HEVC_CTX_PART_MODE + 3)) part_mode_ = hor ? PART_2NxN : PART_Nx2N;
                else { const int b = cb_.bypass(); part_mode_ = hor ? (b ? PART_2NxnD : PART_2NxnU) : (b ? PART_nRx2N : PART_nLx2N); }
            }
        }
    }
    {   // per-4x4 maps of the coding unit, a row of units at a time (the unit loop with its eight array stores was 7 % of the parse)
        const int nu = n >> 2;
        HevcMotion blank; memset(&blank, 0, sizeof blank); blank.ref[0] = blank.ref[1] = -1;
        uint8_t *pm = pm_.data(), *sk = skip_.data(), *nf = nofilter_.data(), *ip = ipm_.data();
        HevcMotion *mo = mot_.data();
        for (int r = 0; r < nu; r++) {
            const int i = i4(x0, y0 + 4 * r);
            const uint8_t vpm = cu_intra_ ? 2 : 1, vsk = cu_skip_, vnf = tq_bypass_;
            fill_units(pm + i, vpm, nu); fill_units(sk + i, vsk, nu); fill_units(nf + i, vnf, nu); fill_units(ip + i, 1, nu);
            if (cu_intra_) for (int k = 0; k < nu; k++) mo[i + k] = blank;      // (the prediction units of an inter unit cover it and write their own)
        }
    }
    if (dg_->on) { dg(0x4000 | (cu_skip_ << 8) | (cu_intra_ << 7) | (tq_bypass_ << 6) | (part_mode_ << 3) | log2); dg(x0); dg(y0); }
    bool pcm = false, root_cbf = true;
    // a prediction unit that gives up (reference index out of range, data overrun) leaves part of the unit without motion of its own: blank the whole
    // unit, so that the picture -- which is still submitted, with its error counted -- carries no stale vectors or unchecked reference indices into
    // the strengths, the collocated field or the motion jobs
    auto drop_inter_cu = [&]() {
        HevcMotion none; memset(&none, 0, sizeof none); none.ref[0] = none.ref[1] = -1;
        for (int r = 0; r < (n >> 2); r++) for (int k = 0; k < (n >> 2); k++) { const int i = i4(x0, y0 + 4 * r) + k; mot_[i] = none; pm_[i] = 2; }
        return false;
    };
    if (cu_skip_) { if (!prediction_unit(x0, y0, n, x0, y0, n, n, 0)) return drop_inter_cu(); }
    else if (cu_intra_) {
        jobs_->n_intra_cu++;
        if (part_mode_ == PART_2Nx2N && sps_->pcm && log2 >= sps_->log2_min_pcm && log2 <= sps_->log2_max_pcm) pcm = cb_.terminate();
        if (pcm) {
            // 9.3.2.5: the arithmetic decoder has consumed exactly the encoder's flush; the samples start at the next byte boundary
            size_t pos = (cb_.bits_consumed() + 7) >> 3;
            const uint8_t *p = cb_.start + pos;
            BitReader br(p < cb_.end ? p : cb_.end, p < cb_.end ? (size_t)(cb_.end - p) : 0);
            for (int c = 0; c < 3; c++) {
                const int sc = c ? 1 : 0, nn = n >> sc, bits = c ? sps_->pcm_bits_c : sps_->pcm_bits_y, lg = log2 - sc;
                HevcIntraTb t; memset(&t, 0, sizeof t);
                t.x = (uint16_t)(x0 >> sc); t.y = (uint16_t)(y0 >> sc); t.log2 = (uint8_t)lg; t.plane = (uint8_t)c; t.mode = kHevcModePcm; t.flags = HTB_BYPASS;
                t.coef_off = (uint32_t)jobs_->coefs.size();
                for (int k = 0; k < nn * nn; k++) { int v = (int)br.u(bits); if (dg_->on) dg(v); v <<= 8 - bits;
                    if (v) jobs_->coefs.push_back((uint32_t)k | ((uint32_t)v << 16)); }
                t.coef_n = (uint32_t)jobs_->coefs.size() - t.coef_off;
                jobs_->itbs.push_back(t);
            }
            if (br.overrun()) return false;
            const uint8_t *next = p + ((br.bitpos() + 7) >> 3);
            if (next > cb_.end) return false;
            { const uint8_t *st = cb_.start; cb_.init_engine(next, cb_.end); (void)st; }
            if (sps_->pcm_loop_filter_disabled) for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) nofilter_[i4(x, y)] = 1;
            root_cbf = false;
        } else {
            const int np = part_mode_ == PART_NxN ? 2 : 1, pb = n / np;
            intra_split_ = np == 2;
            int prev_flag[4], modes[4];
            for (int k = 0; k < np * np; k++) prev_flag[k] = cb_.decision(HEVC_CTX_PREV_INTRA);
            for (int k = 0; k < np * np; k++) {
                const int xp = x0 + (k & 1) * pb, yp = y0 + (k >> 1) * pb;
                int idx;
                if (prev_flag[k]) { idx = 0; while (idx < 2 && cb_.bypass()) idx++; } else { idx = 0;
                    for (int b = 0; b < 5; b++) idx = (idx << 1) | cb_.bypass(); }
                int a = 1, b = 1;                                       // 8.4.2
                if (avail_zs(xp, yp, xp - 1, yp) && pm_[i4(xp - 1, yp)] == 2) a = ipm_[i4(xp - 1, yp)];
                if (((yp - 1) >> sps_->log2_ctb) == (yp >> sps_->log2_ctb) && avail_zs(xp, yp, xp, yp - 1) && pm_[i4(xp, yp - 1)] == 2) b = ipm_[i4(xp,
                    yp - 1)];
                int cand[3];
                if (a == b) { if (a < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; } else { cand[0] = a; cand[1] = 2 + ((a + 29) & 31);
                    cand[2] = 2 + ((a - 1) & 31); } }
                else { cand[0] = a; cand[1] = b; cand[2] = (a && b) ? 0 : ((a != 1 && b != 1) ? 1 : 26); }
                int mode;
                if (prev_flag[k]) mode = cand[idx];
                else { std::sort(cand, cand + 3); mode = idx; for (int i = 0; i < 3; i++) if (mode >= cand[i]) mode++; }
                modes[k] = mode;
                for (int y = yp; y < yp + pb; y += 4) for (int x = xp; x < xp + pb; x += 4) ipm_[i4(x, y)] = (uint8_t)mode;
            }
            int cm = 4;
            if (cb_.decision(HEVC_CTX_INTRA_CHROMA)) { cm = cb_.bypass(); cm = (cm << 1) | cb_.bypass(); }
            static const uint8_t ctab[4] = {0, 26, 10, 1};
            ipm_c_ = cm == 4 ? modes[0] : (ctab[cm] == modes[0] ? 34 : ctab[cm]);
            if (dg_->on) { for (int k = 0; k < np * np; k++) dg(modes[k]); dg(ipm_c_); }
        }
    } else {
        int w[4], h[4], xs[4], ys[4], np = 2;
        xs[0] = x0; ys[0] = y0;
        switch (part_mode_) {
        case PART_2Nx2N: np = 1; w[0] = h[0] = n; break;
        case PART_2NxN: w[0] = w[1] = n; h[0] = h[1] = n / 2; xs[1] = x0; ys[1] = y0 + n / 2; break;
        case PART_Nx2N: w[0] = w[1] = n / 2; h[0] = h[1] = n; xs[1] = x0 + n / 2; ys[1] = y0; break;
        case PART_2NxnU: w[0] = w[1] = n; h[0] = n / 4; h[1] = 3 * n / 4; xs[1] = x0; ys[1] = y0 + n / 4; break;
        case PART_2NxnD: w[0] = w[1] = n; h[0] = 3 * n / 4; h[1] = n / 4; xs[1] = x0; ys[1] = y0 + 3 * n / 4; break;
        case PART_nLx2N: h[0] = h[1] = n; w[0] = n / 4; w[1] = 3 * n / 4; xs[1] = x0 + n / 4; ys[1] = y0; break;
        case PART_nRx2N: h[0] = h[1] = n; w[0] = 3 * n / 4; w[1] = n / 4; xs[1] = x0 + 3 * n / 4; ys[1] = y0; break;
        default: np = 4; for (int k = 0; k < 4; k++) { w[k] = h[k] = n / 2; xs[k] = x0 + (k & 1) * n / 2; ys[k] = y0 + (k >> 1) * n / 2; } break;
        }
        for (int k = 0; k < np; k++) if (!prediction_unit(x0, y0, n, xs[k], ys[k], w[k], h[k], k)) return drop_inter_cu();
    }
    if (cb_.overrun) return false;
    if (!pcm && !cu_skip_) {
        if (!cu_intra_ && !(part_mode_ == PART_2Nx2N && last_merge_)) root_cbf = cb_.decision(HEVC_CTX_RQT_ROOT_CBF);
        if (root_cbf) {
            max_tr_depth_ = cu_intra_ ? sps_->depth_intra + (intra_split_ ? 1 : 0) : sps_->depth_inter;
            if (!transform_tree(x0, y0, x0, y0, log2, 0, 0, 1, 1)) return false;
        }
    }
    for (int r = 0; r < (n >> 2); r++) fill_units(qp_.data() + i4(x0, y0 + 4 * r), (int8_t)qp_y_, n >> 2);
    last_cu_qp_ = qp_y_; cu_since_reset_ = true;
    if (dg_->on) dg(0x4800 | qp_y_);
    return !cb_.overrun;
}

// 7.3.8.4
// HevcCtb.intra_edge: does an intra block of the CTB reach the CTB's (or the picture's) bottom row / right column?  Only then do the CTBs below / the CTB to
// the right read samples that k_hevc_intra reconstructs here -- everything else they see of this CTB was final before that kernel started, and they need
// not wait for it.
void HevcPicParser::note_intra_bottom(int rs) {
    HevcCtb &cj = jobs_->ctbs[rs];
    const int y_end = std::min(h_, ((rs / ctb_w_) + 1) << sps_->log2_ctb), x_end = std::min(w_, ((rs % ctb_w_) + 1) << sps_->log2_ctb);
    cj.intra_edge = 0;
    for (uint32_t i = 0; i < cj.intra_count && cj.intra_edge != 3; i++) {
        const HevcIntraTb &t = jobs_->itbs[cj.intra_first + i];
        const int sc = t.plane ? 1 : 0;
        if ((int)t.y + (1 << t.log2) >= (y_end >> sc)) cj.intra_edge |= 1;
        if ((int)t.x + (1 << t.log2) >= (x_end >> sc)) cj.intra_edge |= 2;
    }
}

bool HevcPicParser::coding_quadtree(int x0, int y0, int log2, int depth) {
    const int n = 1 << log2;
    bool split;
    if (x0 + n <= w_ && y0 + n <= h_ && log2 > sps_->log2_min_cb) {
        const int inc = (avail_zs(x0, y0, x0 - 1, y0) && depth_[i4(x0 - 1, y0)] > depth) + (avail_zs(x0, y0, x0, y0 - 1) && depth_[i4(x0, y0 - 1)] > depth);
        split = cb_.decision(HEVC_CTX_SPLIT_CU + inc);
    } else split = log2 > sps_->log2_min_cb;
    if (pps_->cu_qp_delta && log2 >= sps_->log2_ctb - pps_->diff_cu_qp_delta_depth) {       // a quantisation group starts
        dqp_coded_ = false; dqp_ = 0;
        if (cu_since_reset_) { qp_prev_ = last_cu_qp_; first_qg_ = false; }
    }
    if (split) {
        const int hh = n >> 1;
        for (int k = 0; k < 4; k++) { const int x = x0 + (k & 1) * hh, y = y0 + (k >> 1) * hh;
            if (x < w_ && y < h_ && !coding_quadtree(x, y, log2 - 1, depth + 1)) return false; }
        return true;
    }
    for (int r = 0; r < (n >> 2); r++) fill_units(depth_.data() + i4(x0, y0 + 4 * r), depth, n >> 2);
    derive_qp(x0, y0);
    return coding_unit(x0, y0, log2);
}

// ------------------------------------------------------------------------------------------------------------
// 7.3.8.1 slice_segment_data()
std::string HevcPicParser::parse_slice(const HevcSliceHeader &sh, const HevcSliceRefs &refs, const uint8_t *rbsp, size_t len) {
    if (err_) return "tile layout does not cover the picture";
    sh_ = &sh; refs_ = &refs;
    if (sh.data_offset >= len) return "slice segment without data";
    const int n_ctb = ctb_w_ * ctb_h_;
    if (!sh.dependent) {
        SliceInfo si; memset(&si, 0, sizeof si);
        si.addr = sh.slice_addr; si.deblock_disabled = sh.deblock_disabled; si.lf_across = sh.lf_across_slices; si.beta_off = (int8_t)sh.beta_off;
        si.tc_off = (int8_t)sh.tc_off;
        memcpy(si.slot, refs.slot, sizeof si.slot); memcpy(si.poc, refs.poc, sizeof si.poc); memcpy(si.is_lt, refs.is_lt, sizeof si.is_lt);
        slices_.push_back(si);
        wp_index_ = 0;
        if (sh.has_wp) {
            HevcWp wp; memset(&wp, 0, sizeof wp);
            wp.log2wd[0] = (int16_t)(sh.wp_denom[0] + 6); wp.log2wd[1] = (int16_t)(sh.wp_denom[1] + 6);
            memcpy(wp.w, sh.wp_w, sizeof wp.w); memcpy(wp.o, sh.wp_o, sizeof wp.o);
            jobs_->wps.push_back(wp); wp_index_ = (uint16_t)jobs_->wps.size();
        }
    } else if (slices_.empty()) return "dependent slice segment without a slice";
    slice_idx_ = (int)slices_.size() - 1;
    if (!sh.deblock_disabled) jobs_->any_deblock = true;
    ctb_rs_ = sh.segment_addr; ctb_ts_ = rs2ts_[ctb_rs_]; nbf_rs_ = -1;
    if (sh.dependent) {
        if (!dep_valid_) return "dependent slice segment without stored context variables";
        memcpy(cb_.state, dep_state_, HEVC_N_CTX * sizeof(Cabac::State));
        qp_prev_ = last_cu_qp_; first_qg_ = false; cu_since_reset_ = false;
    } else { init_contexts(); first_qg_ = true; cu_since_reset_ = false; qp_prev_ = sh.qp; }
    qp_y_ = sh.qp; dqp_ = 0; dqp_coded_ = false;
    cb_.init_engine(rbsp + sh.data_offset, rbsp + len);
    bool first_ctu = true;
    for (;;) {
        const int rx = ctb_rs_ % ctb_w_, ry = ctb_rs_ / ctb_w_, tile = tile_id_[ctb_ts_];
        const bool first_in_tile = ctb_ts_ == 0 || tile_id_[ctb_ts_ - 1] != tile;
        const bool row_start = pps_->wpp && (rx == 0 || tile_id_[rs2ts_[ctb_rs_ - 1]] != tile);
        // the collocated picture's motion down to this CTB row (its own parse sizes and fills the field: never read it before)
        if (refs.col && (rx == 0 || first_ctu)) refs.col->wait_rows(std::min((h_ + 15) >> 4, ((ry + 1) << sps_->log2_ctb) >> 4));
        if (ctb_slice_[ctb_rs_] >= 0) return "coding tree block decoded twice";
        ctb_slice_[ctb_rs_] = sh.slice_addr; ctb_sidx_[ctb_rs_] = (uint16_t)slice_idx_;
        { HevcCtb &cj = jobs_->ctbs[ctb_rs_]; cj.beta_off = slices_[slice_idx_].beta_off; cj.tc_off = slices_[slice_idx_].tc_off;
            cj.intra_first = (uint32_t)jobs_->itbs.size(); }
        // 9.3.1: the first CTB of a tile ALWAYS starts from initialised context variables -- also when it opens a dependent slice segment (the
        // stored variables of the previous segment only apply to a segment that starts inside a tile)
        if (first_in_tile) { if (!first_ctu || sh.dependent) init_contexts(); first_qg_ = true; cu_since_reset_ = false; qp_prev_ = sh.qp; }
        else if (row_start) {                                           // 9.3.1: synchronisation with the CTB above and to the right
            const int x0 = rx << sps_->log2_ctb, y0 = ry << sps_->log2_ctb;
            if (wpp_valid_ && avail_zs(x0, y0, x0 + ctb_size_, y0 - ctb_size_)) memcpy(cb_.state, wpp_state_, HEVC_N_CTX * sizeof(Cabac::State));
            else if (!first_ctu) init_contexts();
            first_qg_ = true; cu_since_reset_ = false; qp_prev_ = sh.qp;
        }
        first_ctu = false;
        parse_sao(ctb_rs_);
        if (!coding_quadtree(rx << sps_->log2_ctb, ry << sps_->log2_ctb, sps_->log2_ctb, 0) || cb_.overrun) return "corrupt slice data";
        jobs_->ctbs[ctb_rs_].intra_count = (uint32_t)jobs_->itbs.size() - jobs_->ctbs[ctb_rs_].intra_first;
        note_intra_bottom(ctb_rs_);
        if (pps_->wpp && (rx == 1 || (ctb_rs_ > 1 && rx > 1 && tile_id_[rs2ts_[ctb_rs_ - 2]] != tile))) {
            memcpy(wpp_state_, cb_.state, HEVC_N_CTX * sizeof(Cabac::State)); wpp_valid_ = true; }
        // a finished CTB row (pictures without tiles: rows complete in order) makes its part of the motion field final
        if (col_out_ && !pps_->tiles && rx == ctb_w_ - 1 && ry * (ctb_size_ >> 4) == exported_rows_) export_motion_rows(exported_rows_, std::min(col_out_->h16,
            (ry + 1) * (ctb_size_ >> 4)));
        const int end = cb_.terminate();
        ctb_ts_++;
        if (end) break;
        if (ctb_ts_ >= n_ctb) return "slice data runs past the last coding tree block";
        ctb_rs_ = ts2rs_[ctb_ts_];
        const bool new_tile = pps_->tiles && tile_id_[ctb_ts_] != tile_id_[ctb_ts_ - 1];
        const bool new_row = pps_->wpp && (ctb_rs_ % ctb_w_ == 0 || tile_id_[ctb_ts_] != tile_id_[rs2ts_[ctb_rs_ - 1]]);
        if (new_tile || new_row) {
            if (!cb_.terminate()) return "end_of_subset_one_bit missing";
            const uint8_t *next = cb_.start + ((cb_.bits_consumed() + 7) >> 3);
            if (next >= cb_.end) return "substream missing";
            cb_.init_engine(next, cb_.end);
        }
    }
    if (pps_->dependent_slices) { memcpy(dep_state_, cb_.state, HEVC_N_CTX * sizeof(Cabac::State)); dep_valid_ = true; }
    return cb_.overrun ? "corrupt slice data" : "";
}

// ------------------------------------------------------------------------------------------------------------
// picture end: 8.7.2.2 - 8.7.2.4 edge flags and boundary strengths, QP map, SAO neighbour masks, motion for temporal prediction
void HevcPicParser::export_motion_rows(int r0, int r1) {
    HevcColMotion &c = *col_out_;
    for (int y = r0; y < r1; y++) for (int x = 0; x < c.w16; x++) {
        const int i = i4(x * 16, y * 16); const size_t e = (size_t)y * c.w16 + x;
        c.intra[e] = pm_[i] != 1; c.mot[e] = mot_[i]; c.lt[e] = 0;
        const SliceInfo &s = slices_[ctb_sidx_[((y * 16) >> sps_->log2_ctb) * ctb_w_ + ((x * 16) >> sps_->log2_ctb)]];
        for (int l = 0; l < 2; l++) if (pm_[i] == 1 && ((mot_[i].pf >> l) & 1)) { c.ref_poc[2 * e + l] = s.poc[l][mot_[i].ref[l]];
            c.lt[e] |= (uint8_t)(s.is_lt[l][mot_[i].ref[l]] << l); }
    }
    exported_rows_ = r1;
    c.publish_rows(r1);
}

void HevcPicParser::finish_picture() {
    HevcColMotion *col = col_out_;
    const int lc = sps_->log2_ctb;
    // coding tree blocks no slice delivered (lost / damaged slice segments): the colocated block of the first list-0 reference (zero motion, no
    // residual), flat grey in a picture without references; exempt from filtering
    int conceal_slot = -1;
    for (const SliceInfo &si : slices_) if (si.slot[0][0] >= 0) { conceal_slot = si.slot[0][0]; break; }
    for (int rs = 0; rs < ctb_w_ * ctb_h_; rs++) if (ctb_slice_[rs] < 0) {
        const int x0 = (rs % ctb_w_) << lc, y0 = (rs / ctb_w_) << lc;
        HevcCtb &cj = jobs_->ctbs[rs]; cj.intra_first = (uint32_t)jobs_->itbs.size();
        if (conceal_slot >= 0) {
            HevcPu j; memset(&j, 0, sizeof j); j.slot0 = (int8_t)conceal_slot; j.slot1 = -1;
            for (int y = y0; y < std::min(h_, y0 + ctb_size_); y += 16) for (int x = x0; x < std::min(w_, x0 + ctb_size_); x += 16) {
                j.x = (uint16_t)x; j.y = (uint16_t)y; j.w = (uint8_t)std::min(16, w_ - x); j.h = (uint8_t)std::min(16, h_ - y);
                jobs_->pus.push_back(j);
            }
        } else
        for (int c = 0; c < 3; c++) for (int y = y0; y < std::min(h_, y0 + ctb_size_); y += 8) for (int x = x0; x < std::min(w_, x0 + ctb_size_); x += 8) {
            // 8x8 luma / 4x4 chroma blocks of the value 128 (PCM-style: no prediction, the "coefficients" are the samples)
            const int sc = c ? 1 : 0, lg = 3 - sc, nn = 1 << lg;
            HevcIntraTb t; memset(&t, 0, sizeof t);
            t.x = (uint16_t)(x >> sc); t.y = (uint16_t)(y >> sc); t.log2 = (uint8_t)lg; t.plane = (uint8_t)c; t.mode = kHevcModePcm; t.flags = HTB_BYPASS;
            t.coef_off = (uint32_t)jobs_->coefs.size();
            for (int k = 0; k < nn * nn; k++) jobs_->coefs.push_back((uint32_t)k | (128u << 16));
            t.coef_n = (uint32_t)(nn * nn);
            jobs_->itbs.push_back(t);
        }
        cj.intra_count = (uint32_t)jobs_->itbs.size() - cj.intra_first;
        note_intra_bottom(rs);
        for (int y = y0; y < std::min(h_, y0 + ctb_size_); y += 4) for (int x = x0; x < std::min(w_, x0 + ctb_size_); x += 4) { const int i = i4(x, y);
            pm_[i] = 2; nofilter_[i] = 1; qp_[i] = 26; }
    }
    if (slices_.empty()) { SliceInfo si; memset(&si, 0, sizeof si); si.deblock_disabled = true; slices_.push_back(si); }
    const int w8 = w_ >> 3, h8 = h_ >> 3;
    jobs_->qp8.resize((size_t)w8 * h8);
    for (int y = 0; y < h8; y++) for (int x = 0; x < w8; x++) { const int i = i4(x * 8, y * 8);
        jobs_->qp8[(size_t)y * w8 + x] = (uint8_t)((qp_[i] & 63) | (nofilter_[i] ? 128 : 0)); }
    // Deblocking (8.7.2): the boundary strengths are derived on the device (k_hevc_bs_raster / k_hevc_bs, round 5) from the prediction blocks, intra blocks
    // and coded transform blocks of the job lists; what they cannot know -- the slice's deblocking switch, slice / tile boundaries that must not be filtered
    // across (8.7.2.3: the left / top edge of a slice takes the flag of the slice that holds q0), concealed CTBs -- goes into HevcCtb.db_flags.
    if (jobs_->any_deblock) {
        for (int rs = 0; rs < ctb_w_ * ctb_h_; rs++) {
            const SliceInfo &sq = slices_[ctb_sidx_[rs]];
            uint8_t f = 0;
            if (sq.deblock_disabled) f |= HDB_DISABLED;
            if (ctb_slice_[rs] < 0) f |= HDB_CONCEALED;
            auto blocked = [&](int nrs) {
                const SliceInfo &sp = slices_[ctb_sidx_[nrs]];
                if (sq.addr != sp.addr && !sq.lf_across) return true;
                return !pps_->lf_across_tiles && tile_id_[rs2ts_[rs]] != tile_id_[rs2ts_[nrs]];
            };
            if (rs % ctb_w_ > 0 && blocked(rs - 1)) f |= HDB_NO_LEFT;
            if (rs >= ctb_w_ && blocked(rs - ctb_w_)) f |= HDB_NO_TOP;
            jobs_->ctbs[rs].db_flags = f;
        }
    }
    // SAO: which neighbouring CTBs the edge offset of a CTB may read (8.7.3: slice and tile boundaries)
    if (jobs_->any_sao) {
        static const int dx[8] = {-1, 1, 0, 0, -1, 1, -1, 1}, dy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
        for (int rs = 0; rs < ctb_w_ * ctb_h_; rs++) {
            const int cx = rs % ctb_w_, cy = rs / ctb_w_; uint8_t mask = 0;
            const SliceInfo &sc = slices_[ctb_sidx_[rs]];
            for (int k = 0; k < 8; k++) {
                const int nx = cx + dx[k], ny = cy + dy[k];
                if (nx < 0 || ny < 0 || nx >= ctb_w_ || ny >= ctb_h_) continue;
                const int nrs = ny * ctb_w_ + nx;
                const SliceInfo &sn = slices_[ctb_sidx_[nrs]];
                bool ok = true;
                if (sn.addr != sc.addr) ok = rs2ts_[nrs] < rs2ts_[rs] ? sc.lf_across : sn.lf_across;
                if (ok && !pps_->lf_across_tiles && tile_id_[rs2ts_[nrs]] != tile_id_[rs2ts_[rs]]) ok = false;
                if (ok) mask |= (uint8_t)(1 << k);
            }
            jobs_->ctbs[rs].nb_mask = mask;
        }
    }
    if (col && exported_rows_ < col->h16) export_motion_rows(exported_rows_, col->h16);      // what the row-wise export did not cover (tiles, damaged pictures)
}

}  // namespace jmamd
