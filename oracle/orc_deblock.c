/*
 * oracle/orc_deblock.c -- CPU ORACLE (test infrastructure only).
 * In-loop deblocking filter, H.264 clause 8.7, applied macroblock by
 * macroblock in raster order (vertical edges left->right, then horizontal
 * edges top->bottom, per MB) exactly as the clause orders it.  Part of the
 * closed cuvidDecodePicture reconstruction (/root/reference/nv_dec/nv_dec.cpp:37).
 */
#include "orc_internal.h"
#include "orc_tables.h"

/* 8.7.2.1 boundary strength between 4x4 luma blocks p (in MB mp, raster bp)
 * and q (in MB mq, raster bq); mb_edge = edge lies on a macroblock boundary.
 * field = the picture is a field picture, horizontal = the edge is a horizontal one: the value 4 is for frame macroblocks and for the vertical edges
 * of field macroblocks -- an intra macroblock edge that runs horizontally through a field gets 3; and the vertical vector difference that counts as
 * "one frame sample" is 2 quarter FIELD samples. */
static int bs_of(OrcDec *d, const MbInfo *mp, int bp, const MbInfo *mq, int bq, int mb_edge, int field, int horizontal) {
    if (mp->is_intra || mq->is_intra) { if (mb_edge && field && horizontal) d->stats[ORC_ST_FIELD_BS3]++; return mb_edge && !(field && horizontal) ? 4 : 3; }
    const int vlimit = field ? 2 : 4;
    int nzp, nzq;
    if (mp->t8x8) { int o = ((bp >> 3) * 2 + ((bp & 3) >> 1)); o = (o >> 1) * 8 + (o & 1) * 2;
        nzp = mp->total_coeff[o] | mp->total_coeff[o + 1] | mp->total_coeff[o + 4] | mp->total_coeff[o + 5]; }
    else nzp = mp->total_coeff[bp];
    if (mq->t8x8) { int o = ((bq >> 3) * 2 + ((bq & 3) >> 1)); o = (o >> 1) * 8 + (o & 1) * 2;
        nzq = mq->total_coeff[o] | mq->total_coeff[o + 1] | mq->total_coeff[o + 4] | mq->total_coeff[o + 5]; }
    else nzq = mq->total_coeff[bq];
    if (nzp || nzq) return 2;
    /* different reference pictures or a different number of motion vectors; the lists themselves do not matter */
    int pb8 = (bp >> 3) * 2 + ((bp & 3) >> 1), qb8 = (bq >> 3) * 2 + ((bq & 3) >> 1);
    int p0 = mp->ref_idx[0][pb8] >= 0 ? mp->ref_pic_id[0][pb8] : -1, p1 = mp->ref_idx[1][pb8] >= 0 ? mp->ref_pic_id[1][pb8] : -1;
    int q0 = mq->ref_idx[0][qb8] >= 0 ? mq->ref_pic_id[0][qb8] : -1, q1 = mq->ref_idx[1][qb8] >= 0 ? mq->ref_pic_id[1][qb8] : -1;
    const int16_t *pm0 = mp->mv[0][bp], *pm1 = mp->mv[1][bp], *qm0 = mq->mv[0][bq], *qm1 = mq->mv[1][bq];
#define FAR(a, b) (orc_abs((a)[0] - (b)[0]) >= 4 || (orc_abs((a)[1] - (b)[1]) >= vlimit && (orc_abs((a)[1] - (b)[1]) >= 4 || ++d->stats[ORC_ST_FIELD_MVY])))
    int np = (p0 >= 0) + (p1 >= 0), nq = (q0 >= 0) + (q1 >= 0);
    if (np != nq) return 1;
    if (np == 1) {
        int rp = p0 >= 0 ? p0 : p1, rq = q0 >= 0 ? q0 : q1;
        if (rp != rq) return 1;
        return FAR(p0 >= 0 ? pm0 : pm1, q0 >= 0 ? qm0 : qm1);
    }
    if (!((p0 == q0 && p1 == q1) || (p0 == q1 && p1 == q0))) return 1;
    if (p0 != p1) {
        if (p0 == q0) return FAR(pm0, qm0) || FAR(pm1, qm1);
        return FAR(pm0, qm1) || FAR(pm1, qm0);
    }
    return (FAR(pm0, qm0) || FAR(pm1, qm1)) && (FAR(pm0, qm1) || FAR(pm1, qm0));
#undef FAR
}

/* 8.7.2.3 / 8.7.2.4: filter one line of samples across an edge.
 * q points at q0; step = distance between successive samples across the edge */
static void filter_line(uint8_t *q, int step, int bS, int alpha, int beta, int indexA, int chroma) {
    int p0 = q[-step], p1 = q[-2 * step], q0 = q[0], q1 = q[step];
    if (!(orc_abs(p0 - q0) < alpha && orc_abs(p1 - p0) < beta && orc_abs(q1 - q0) < beta)) return;
    if (chroma) {
        if (bS < 4) {
            int tc = orc_tc0[indexA][bS - 1] + 1;
            int delta = orc_clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
            q[-step] = (uint8_t)orc_clip1(p0 + delta); q[0] = (uint8_t)orc_clip1(q0 - delta);
        } else {
            q[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2); q[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
        }
        return;
    }
    int p2 = q[-3 * step], q2 = q[2 * step];
    int ap = orc_abs(p2 - p0), aq = orc_abs(q2 - q0);
    if (bS < 4) {
        int tc0 = orc_tc0[indexA][bS - 1];
        int tc = tc0 + (ap < beta) + (aq < beta);
        int delta = orc_clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        q[-step] = (uint8_t)orc_clip1(p0 + delta); q[0] = (uint8_t)orc_clip1(q0 - delta);
        if (ap < beta) q[-2 * step] = (uint8_t)(p1 + orc_clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1));
        if (aq < beta) q[step] = (uint8_t)(q1 + orc_clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1));
    } else {
        int p3 = q[-4 * step], q3 = q[3 * step];
        int strong = orc_abs(p0 - q0) < ((alpha >> 2) + 2);
        if (ap < beta && strong) {
            q[-step] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            q[-2 * step] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
            q[-3 * step] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
        } else q[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        if (aq < beta && strong) {
            q[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            q[step] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
            q[2 * step] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        } else q[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    }
}

void orc_deblock_picture(OrcDec *d, Picture *pic) {
    int mbw = d->mb_w, mbh = d->mb_h;
    for (int my = 0; my < mbh; my++) for (int mx = 0; mx < mbw; mx++) {
        MbInfo *mq = &pic->mbs[my * mbw + mx];
        if (mq->slice_num < 0 || mq->disable_deblock == 1) continue;
        for (int dir = 0; dir < 2; dir++) {              /* 0: vertical edges, 1: horizontal edges */
            for (int e = 0; e < 4; e++) {
                const MbInfo *mp = mq;
                if (e == 0) {
                    int nx = mx - (dir == 0), ny = my - (dir == 1);
                    if (nx < 0 || ny < 0) continue;
                    mp = &pic->mbs[ny * mbw + nx];
                    if (mp->slice_num < 0) continue;
                    if (mq->disable_deblock == 2 && mp->slice_num != mq->slice_num) continue;
                } else if (mq->t8x8 && (e & 1)) continue;
                int bS[4];
                for (int k = 0; k < 4; k++) {
                    int bq = dir == 0 ? k * 4 + e : e * 4 + k;
                    int bp = e == 0 ? (dir == 0 ? k * 4 + 3 : 12 + k) : (dir == 0 ? bq - 1 : bq - 4);
                    bS[k] = bs_of(d, mp, bp, mq, bq, e == 0, pic->is_field, dir == 1);
                }
                if (!(bS[0] | bS[1] | bS[2] | bS[3])) continue;
                /* luma */
                {
                    int qpav = (mp->qp + mq->qp + 1) >> 1;
                    int ia = orc_clip3(0, 51, qpav + mq->alpha_off), ib = orc_clip3(0, 51, qpav + mq->beta_off);
                    int alpha = orc_alpha[ia], beta = orc_beta[ib];
                    for (int i = 0; i < 16; i++) {
                        if (!bS[i >> 2]) continue;
                        uint8_t *q = dir == 0 ? pic->y + (my * 16 + i) * pic->stride_y + mx * 16 + e * 4
                                              : pic->y + (my * 16 + e * 4) * pic->stride_y + mx * 16 + i;
                        filter_line(q, dir == 0 ? 1 : pic->stride_y, bS[i >> 2], alpha, beta, ia, 0);
                    }
                }
                /* chroma: edges 0 and 2 of the luma grid map to chroma edges 0 and 4 */
                if (!(e & 1)) for (int pl = 0; pl < 2; pl++) {
                    uint8_t *base = pl ? pic->v : pic->u;
                    int qpav = (mp->qpc[pl] + mq->qpc[pl] + 1) >> 1;
                    int ia = orc_clip3(0, 51, qpav + mq->alpha_off), ib = orc_clip3(0, 51, qpav + mq->beta_off);
                    int alpha = orc_alpha[ia], beta = orc_beta[ib];
                    for (int i = 0; i < 8; i++) {
                        if (!bS[i >> 1]) continue;
                        uint8_t *q = dir == 0 ? base + (my * 8 + i) * pic->stride_c + mx * 8 + e * 2
                                              : base + (my * 8 + e * 2) * pic->stride_c + mx * 8 + i;
                        filter_line(q, dir == 0 ? 1 : pic->stride_c, bS[i >> 1], alpha, beta, ia, 1);
                    }
                }
            }
        }
    }
}
