/*
 * oracle/orc_hevc_filter.c -- CPU ORACLE (test infrastructure only): HEVC in-loop filters, picture level.
 * Deblocking (ITU-T H.265 8.7.2) and sample adaptive offset (8.7.3); the tail of what cuvidDecodePicture does
 * for HEVC (/root/reference/nv_dec/nv_dec.cpp:33-41, SURVEY.md 8a row a6 "deblock (+SAO for HEVC)").
 */
#include "orc_hevc_internal.h"

#define I4(d, x, y) (((y) >> 2) * (d)->w4 + ((x) >> 2))

static const HSlice *slice_at(const OrchDec *d, int x, int y) { return &d->slices[d->slice_of4[I4(d, x, y)]]; }

/* 8.7.2.4 boundary filtering strength of the 4-sample edge segment whose first q sample is (xq, yq); dir 0 = vertical edge */
static int edge_bs(const OrchDec *d, int xq, int yq, int dir) {
    int xp = dir ? xq : xq - 1, yp = dir ? yq - 1 : yq;
    int iq = I4(d, xq, yq), ip = I4(d, xp, yp);
    int tu_edge = d->edge[iq] & (dir ? 2 : 1), pu_edge = d->edge[iq] & (dir ? 8 : 4);
    if (!tu_edge && !pu_edge) return 0;
    const HSlice *sq = slice_at(d, xq, yq), *sp = slice_at(d, xp, yp);
    if (sq->deblock_disabled) return 0;
    /* 8.7.2.3: left / top edge of the slice or tile the current (q) block belongs to */
    if (sq->slice_addr != sp->slice_addr && !sq->lf_across_slices) return 0;
    {
        int lc = d->asps->log2_ctb;
        int cq = (yq >> lc) * d->ctb_w + (xq >> lc), cp = (yp >> lc) * d->ctb_w + (xp >> lc);
        if (d->tile_id[d->ctb_rs2ts[cq]] != d->tile_id[d->ctb_rs2ts[cp]] && !d->apps->lf_across_tiles) return 0;
    }
    if (d->pred_mode[iq] == 2 || d->pred_mode[ip] == 2) return 2;
    if (tu_edge && (d->cbf[iq] || d->cbf[ip])) return 1;
    const HMotion *mq = &d->mot[iq], *mp = &d->mot[ip];
    int nq = (mq->pred_flag & 1) + (mq->pred_flag >> 1), np = (mp->pred_flag & 1) + (mp->pred_flag >> 1);
    if (nq != np) return 1;
    int rq[2], rp[2]; const int16_t *vq[2], *vp[2]; int k = 0;
    for (int l = 0; l < 2; l++) if (mq->pred_flag >> l & 1) { rq[k] = sq->ref_dpb[l][mq->ref_idx[l]]; vq[k] = mq->mv[l]; k++; }
    k = 0;
    for (int l = 0; l < 2; l++) if (mp->pred_flag >> l & 1) { rp[k] = sp->ref_dpb[l][mp->ref_idx[l]]; vp[k] = mp->mv[l]; k++; }
#define FAR(a, b) (abs((a)[0] - (b)[0]) >= 4 || abs((a)[1] - (b)[1]) >= 4)
    if (nq == 1) return rq[0] != rp[0] || FAR(vq[0], vp[0]);
    if (!((rq[0] == rp[0] && rq[1] == rp[1]) || (rq[0] == rp[1] && rq[1] == rp[0]))) return 1;
    if (rq[0] != rq[1]) {                                                /* two different reference pictures */
        if (rq[0] == rp[0]) return FAR(vq[0], vp[0]) || FAR(vq[1], vp[1]);
        return FAR(vq[0], vp[1]) || FAR(vq[1], vp[0]);
    }
    return (FAR(vq[0], vp[0]) || FAR(vq[1], vp[1])) && (FAR(vq[0], vp[1]) || FAR(vq[1], vp[0]));
#undef FAR
}

/* 8.7.2.5.3 + 8.7.2.5.7: one 4-line luma segment; p points at q0 of line 0, `step` moves across the edge, `line` along it */
static void luma_segment(uint8_t *q0, int step, int line, int bs, int qp, int beta_off, int tc_off, int nofilt_p, int nofilt_q) {
    int beta = orch_beta_tab[h_clip3(0, 51, qp + (beta_off << 1))];
    int tc = orch_tc_tab[h_clip3(0, 53, qp + 2 * (bs - 1) + (tc_off << 1))];
#define P(i, k) q0[(k) * line - ((i) + 1) * step]
#define Q(i, k) q0[(k) * line + (i) * step]
    int dp0 = abs(P(2, 0) - 2 * P(1, 0) + P(0, 0)), dp3 = abs(P(2, 3) - 2 * P(1, 3) + P(0, 3));
    int dq0 = abs(Q(2, 0) - 2 * Q(1, 0) + Q(0, 0)), dq3 = abs(Q(2, 3) - 2 * Q(1, 3) + Q(0, 3));
    int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3;
    if (dpq0 + dpq3 >= beta) return;
    int s0 = 2 * dpq0 < (beta >> 2) && abs(P(3, 0) - P(0, 0)) + abs(Q(0, 0) - Q(3, 0)) < (beta >> 3) && abs(P(0, 0) - Q(0, 0)) < ((5 * tc + 1) >> 1);
    int s3 = 2 * dpq3 < (beta >> 2) && abs(P(3, 3) - P(0, 3)) + abs(Q(0, 3) - Q(3, 3)) < (beta >> 3) && abs(P(0, 3) - Q(0, 3)) < ((5 * tc + 1) >> 1);
    int strong = s0 && s3;
    int dep = dp < ((beta + (beta >> 1)) >> 3), deq = dq < ((beta + (beta >> 1)) >> 3);
    for (int k = 0; k < 4; k++) {
        int p0 = P(0, k), p1 = P(1, k), p2 = P(2, k), p3 = P(3, k), q0v = Q(0, k), q1 = Q(1, k), q2 = Q(2, k), q3 = Q(3, k);
        if (strong) {
            if (!nofilt_p) {
                P(0, k) = (uint8_t)h_clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0v + q1 + 4) >> 3);
                P(1, k) = (uint8_t)h_clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0v + 2) >> 2);
                P(2, k) = (uint8_t)h_clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0v + 4) >> 3);
            }
            if (!nofilt_q) {
                Q(0, k) = (uint8_t)h_clip3(q0v - 2 * tc, q0v + 2 * tc, (p1 + 2 * p0 + 2 * q0v + 2 * q1 + q2 + 4) >> 3);
                Q(1, k) = (uint8_t)h_clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0v + q1 + q2 + 2) >> 2);
                Q(2, k) = (uint8_t)h_clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0v + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
            }
        } else {
            int delta = (9 * (q0v - p0) - 3 * (q1 - p1) + 8) >> 4;
            if (abs(delta) >= tc * 10) continue;
            delta = h_clip3(-tc, tc, delta);
            if (!nofilt_p) {
                P(0, k) = (uint8_t)h_clip1(p0 + delta);
                if (dep) P(1, k) = (uint8_t)h_clip1(p1 + h_clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1));
            }
            if (!nofilt_q) {
                Q(0, k) = (uint8_t)h_clip1(q0v - delta);
                if (deq) Q(1, k) = (uint8_t)h_clip1(q1 + h_clip3(-(tc >> 1), tc >> 1, (((q2 + q0v + 1) >> 1) - q1 - delta) >> 1));
            }
        }
    }
#undef P
#undef Q
}
/* 8.7.2.5.5 + 8.7.2.5.8: four chroma lines across one edge */
static void chroma_segment(uint8_t *q0, int step, int line, int qpc, int tc_off, int nofilt_p, int nofilt_q) {
    int tc = orch_tc_tab[h_clip3(0, 53, qpc + 2 + (tc_off << 1))];
    for (int k = 0; k < 4; k++) {
        uint8_t *q = q0 + k * line;
        int p0 = q[-step], p1 = q[-2 * step], q0v = q[0], q1 = q[step];
        int delta = h_clip3(-tc, tc, (((q0v - p0) << 2) + p1 - q1 + 4) >> 3);
        if (!nofilt_p) q[-step] = (uint8_t)h_clip1(p0 + delta);
        if (!nofilt_q) q[0] = (uint8_t)h_clip1(q0v - delta);
    }
}

void orch_deblock_picture(OrchDec *d) {
    HPic *pic = d->cur;
    for (int dir = 0; dir < 2; dir++) {
        /* luma: edges on the 8x8 grid, 4-sample segments */
        for (int y = 0; y < d->h; y += dir ? 8 : 4) for (int x = 0; x < d->w; x += dir ? 4 : 8) {
            if ((dir ? y : x) == 0) continue;
            int bs = edge_bs(d, x, y, dir);
            if (!bs) continue;
            int xp = dir ? x : x - 1, yp = dir ? y - 1 : y;
            const HSlice *sq = slice_at(d, x, y);
            int qp = (d->qp_y[I4(d, x, y)] + d->qp_y[I4(d, xp, yp)] + 1) >> 1;
            luma_segment(pic->pl[0] + y * pic->stride[0] + x, dir ? pic->stride[0] : 1, dir ? 1 : pic->stride[0], bs, qp, sq->beta_offset_div2,
                sq->tc_offset_div2,
                         d->nofilter[I4(d, xp, yp)], d->nofilter[I4(d, x, y)]);
        }
        /* chroma: edges on the 8x8 chroma sample grid, bS == 2 only, four chroma lines per unit (bS taken at the unit's first line) */
        for (int c = 1; c < 3; c++) {
            int cw = d->w >> 1, ch = d->h >> 1, off = c == 1 ? d->apps->cb_qp_offset : d->apps->cr_qp_offset;
            for (int y = 0; y < ch; y += dir ? 8 : 4) for (int x = 0; x < cw; x += dir ? 4 : 8) {
                if ((dir ? y : x) == 0) continue;
                int xl = x * 2, yl = y * 2;
                if (edge_bs(d, xl, yl, dir) != 2) continue;
                int xp = dir ? xl : xl - 1, yp = dir ? yl - 1 : yl;
                const HSlice *sq = slice_at(d, xl, yl);
                int qpi = ((d->qp_y[I4(d, xl, yl)] + d->qp_y[I4(d, xp, yp)] + 1) >> 1) + off;
                chroma_segment(pic->pl[c] + y * pic->stride[c] + x, dir ? pic->stride[c] : 1, dir ? 1 : pic->stride[c], orch_qpc_tab[h_clip3(0, 57, qpi)],
                    sq->tc_offset_div2,
                               d->nofilter[I4(d, xp, yp)], d->nofilter[I4(d, xl, yl)]);
            }
        }
    }
}

/* 8.7.3 sample adaptive offset */
void orch_sao_picture(OrchDec *d) {
    HPic *pic = d->cur;
    if (!d->asps->sao) return;
    int any = 0;
    for (int i = 0; i < d->ctb_w * d->ctb_h; i++) any |= d->sao[i].type[0] | d->sao[i].type[1] | d->sao[i].type[2];
    if (!any) return;
    for (int c = 0; c < 3; c++) { int hh = d->h >> (c ? 1 : 0); memcpy(d->deblocked[c], pic->pl[c], (size_t)pic->stride[c] * (size_t)hh); }
    const int lc = d->asps->log2_ctb;
    for (int c = 0; c < 3; c++) {
        const int sc = c ? 1 : 0, pw = d->w >> sc, ph = d->h >> sc, cs = d->ctb_size >> sc, stride = pic->stride[c];
        const uint8_t *src = d->deblocked[c]; uint8_t *dst = pic->pl[c];
        for (int ry = 0; ry < d->ctb_h; ry++) for (int rx = 0; rx < d->ctb_w; rx++) {
            const HSao *o = &d->sao[ry * d->ctb_w + rx];
            if (!o->type[c]) continue;
            int x0 = rx * cs, y0 = ry * cs;
            int band[32]; memset(band, 0, sizeof band);
            if (o->type[c] == 1) for (int k = 0; k < 4; k++) band[(k + o->band_pos[c]) & 31] = k + 1;
            static const int8_t hp[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}}, vp[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
            for (int y = y0; y < y0 + cs && y < ph; y++) for (int x = x0; x < x0 + cs && x < pw; x++) {
                int xl = x << sc, yl = y << sc;
                if (d->nofilter[I4(d, xl, yl)]) continue;
                int v = src[y * stride + x], offv;
                if (o->type[c] == 1) { int bi = band[v >> 3]; offv = bi ? o->off[c][bi - 1] : 0; }
                else {
                    int cls = o->eo_class[c], e = 2, skip = 0;
                    for (int k = 0; k < 2; k++) {
                        int xn = x + hp[cls][k], yn = y + vp[cls][k];
                        if (xn < 0 || yn < 0 || xn >= pw || yn >= ph) { skip = 1; break; }
                        int xnl = xn << sc, ynl = yn << sc;
                        const HSlice *sn = slice_at(d, xnl, ynl), *scur = slice_at(d, xl, yl);
                        if (sn->slice_addr != scur->slice_addr) {
                            int n_first = d->min_tb_zs[(ynl >> d->asps->log2_min_tb) * d->tb_w + (xnl >> d->asps->log2_min_tb)] <
                                d->min_tb_zs[(yl >> d->asps->log2_min_tb) * d->tb_w + (xl >> d->asps->log2_min_tb)];
                            if (n_first ? !scur->lf_across_slices : !sn->lf_across_slices) { skip = 1; break; }
                        }
                        if (!d->apps->lf_across_tiles) {
                            int cn = (ynl >> lc) * d->ctb_w + (xnl >> lc), cc = (yl >> lc) * d->ctb_w + (xl >> lc);
                            if (d->tile_id[d->ctb_rs2ts[cn]] != d->tile_id[d->ctb_rs2ts[cc]]) { skip = 1; break; }
                        }
                        int nv = src[yn * stride + xn];
                        e += (v > nv) - (v < nv);
                    }
                    if (skip) continue;
                    int idx = e == 2 ? 0 : (e < 2 ? e + 1 : e);          /* edgeIdx: 0 1 2 3 4 -> 1 2 0 3 4 */
                    offv = idx ? o->off[c][idx - 1] : 0;
                }
                dst[y * stride + x] = (uint8_t)h_clip1(v + offv);
            }
        }
    }
}
