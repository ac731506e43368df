/*
 * oracle/orc_cabac.c -- CPU ORACLE (test infrastructure only).
 * CABAC parsing process for slice data, ITU-T H.264 clause 9.3, written from the
 * clause's own steps (initialisation 9.3.1, binarisation 9.3.2, ctxIdx derivation
 * 9.3.3.1, arithmetic decoding engine 9.3.3.2), one bit at a time.  Part of the
 * restatement of the closed picture reconstruction behind cuvidDecodePicture
 * (/root/reference/nv_dec/nv_dec.cpp:37; entropy_coding_mode_flag arrives in
 * CUVIDH264PICPARAMS, nv_sdk/inc/dynlink_cuviddec.h:243-298).  Frame macroblocks only.
 */
#include "orc_slice.h"

/* ------------------------------ 9.3.1.1 ----------------------------------- */
void orc_cabac_init_contexts(Cabac *c, int slice_is_i, int cabac_init_idc, int slice_qp) {
    const int8_t (*mn)[2] = orc_cabac_init_mn[slice_is_i ? 0 : 1 + cabac_init_idc];
    int qp = orc_clip3(0, 51, slice_qp);
    for (int i = 0; i < ORC_CABAC_N_CTX; i++) {
        int pre = orc_clip3(1, 126, ((mn[i][0] * qp) >> 4) + mn[i][1]);
        c->state[i] = pre <= 63 ? (uint8_t)(63 - pre) : (uint8_t)((pre - 64) | 64);
    }
}

/* ------------------------------ 9.3.1.2 ----------------------------------- */
int orc_cabac_init_engine(Cabac *c, Bits *b) {
    c->b = b;
    c->range = 510;
    c->offset = bits_u(b, 9);
    return b->err ? -1 : 0;
}

/* ------------------------------ 9.3.3.2 ----------------------------------- */
static int decision(Cabac *c, int ctx) {
    unsigned st = c->state[ctx] & 63, mps = c->state[ctx] >> 6;
    unsigned lps = orc_cabac_range_lps[st][(c->range >> 6) & 3];
    int bin;
    c->range -= lps;
    if (c->offset >= c->range) {
        bin = !mps; c->offset -= c->range; c->range = lps;
        if (st == 0) mps = 1 - mps;
        st = orc_cabac_trans_lps[st];
    } else {
        bin = (int)mps;
        if (st < 62) st++;
    }
    c->state[ctx] = (uint8_t)(st | (mps << 6));
    while (c->range < 256) { c->range <<= 1; c->offset = (c->offset << 1) | bits_u1(c->b); }   /* RenormD */
    return bin;
}
static int bypass(Cabac *c) {
    c->offset = (c->offset << 1) | bits_u1(c->b);
    if (c->offset >= c->range) { c->offset -= c->range; return 1; }
    return 0;
}
int orc_cabac_terminate(Cabac *c) {
    c->range -= 2;
    if (c->offset >= c->range) return 1;            /* no renormalisation: parsing of the slice / before PCM ends */
    while (c->range < 256) { c->range <<= 1; c->offset = (c->offset << 1) | bits_u1(c->b); }
    return 0;
}

/* --------------------------- neighbour helpers ----------------------------- */
static int mb_is_inxn(const MbInfo *m) { return m->is_intra && !m->is_i16 && !m->is_pcm; }
/* the MB holding the 4x4 block left of / above block (bx,by) of the current MB, and that block's raster index */
static MbInfo *nb4(Sl *s, int bx, int by, int left, int *r) {
    if (left) {
        if (bx > 0) { *r = by * 4 + bx - 1; return s->mb; }
        *r = by * 4 + 3; return orc_sl_mb_at(s, s->mb_x - 1, s->mb_y);
    }
    if (by > 0) { *r = (by - 1) * 4 + bx; return s->mb; }
    *r = 12 + bx; return orc_sl_mb_at(s, s->mb_x, s->mb_y - 1);
}

/* ----------------------------- syntax elements ----------------------------- */
int orc_cabac_mb_skip_flag(Sl *s) {               /* 9.3.3.1.1.1 */
    MbInfo *a = orc_sl_mb_at(s, s->mb_x - 1, s->mb_y), *b = orc_sl_mb_at(s, s->mb_x, s->mb_y - 1);
    int inc = (a && !a->is_skip) + (b && !b->is_skip);
    return decision(&s->c, (s->sh->slice_type == SLICE_B ? 24 : 11) + inc);
}

/* mb_type of an intra macroblock (Table 9-36 I-slice binarisation); base = ctx of bin 0, in_i = I slice */
static int intra_mb_type(Sl *s, int base, int in_i) {
    Cabac *c = &s->c;
    if (in_i) {
        MbInfo *a = orc_sl_mb_at(s, s->mb_x - 1, s->mb_y), *b = orc_sl_mb_at(s, s->mb_x, s->mb_y - 1);
        int inc = (a && !mb_is_inxn(a)) + (b && !mb_is_inxn(b));       /* 9.3.3.1.1.3 */
        if (!decision(c, base + inc)) return 0;                        /* I_NxN */
        base += 2;
    } else if (!decision(c, base)) return 0;
    if (orc_cabac_terminate(c)) return 25;                             /* I_PCM */
    int t = 1;
    t += 12 * decision(c, base + 1);                                   /* CodedBlockPatternLuma != 0 */
    if (decision(c, base + 2)) t += 4 + 4 * decision(c, base + 2 + in_i);   /* chroma 1 / 2 */
    t += 2 * decision(c, base + 3 + in_i);
    t += decision(c, base + 3 + 2 * in_i);
    return t;
}

int orc_cabac_mb_type(Sl *s) {
    Cabac *c = &s->c;
    int st = s->sh->slice_type;
    if (st == SLICE_I) return intra_mb_type(s, 3, 1);
    if (st == SLICE_P) {                                                /* Table 9-37 (a), ctxIdxOffset 14 / suffix 17 */
        if (!decision(c, 14)) {
            if (!decision(c, 15)) return 3 * decision(c, 16);           /* P_L0_16x16 / P_8x8 */
            return 2 - decision(c, 17);                                 /* P_L0_L0_8x16 / P_L0_L0_16x8 */
        }
        return 5 + intra_mb_type(s, 17, 0);
    }
    /* B: Table 9-37 (b), ctxIdxOffset 27 / suffix 32 */
    MbInfo *a = orc_sl_mb_at(s, s->mb_x - 1, s->mb_y), *b = orc_sl_mb_at(s, s->mb_x, s->mb_y - 1);
    int inc = (a && !a->b_direct16) + (b && !b->b_direct16);       /* 9.3.3.1.1.3: B_Skip / B_Direct_16x16 neighbours count 0 */
    if (!decision(c, 27 + inc)) return 0;                               /* B_Direct_16x16 */
    if (!decision(c, 27 + 3)) return 1 + decision(c, 27 + 5);           /* B_L0_16x16 / B_L1_16x16 */
    int bits = decision(c, 27 + 4) << 3;
    bits |= decision(c, 27 + 5) << 2; bits |= decision(c, 27 + 5) << 1; bits |= decision(c, 27 + 5);
    if (bits < 8) return bits + 3;
    if (bits == 13) return 23 + intra_mb_type(s, 32, 0);
    if (bits == 14) return 11;
    if (bits == 15) return 22;
    bits = (bits << 1) | decision(c, 27 + 5);
    return bits - 4;
}

int orc_cabac_sub_mb_type(Sl *s) {
    Cabac *c = &s->c;
    if (s->sh->slice_type == SLICE_P) {                                 /* Table 9-38, ctx 21..23 */
        if (decision(c, 21)) return 0;
        if (!decision(c, 22)) return 1;
        return decision(c, 23) ? 2 : 3;
    }
    if (!decision(c, 36)) return 0;                                     /* ctx 36..39 */
    if (!decision(c, 37)) return 1 + decision(c, 39);
    int t = 3;
    if (decision(c, 38)) {
        if (decision(c, 39)) return 11 + decision(c, 39);
        t += 4;
    }
    t += 2 * decision(c, 39);
    t += decision(c, 39);
    return t;
}

int orc_cabac_transform8x8_flag(Sl *s) {          /* 9.3.3.1.1.10 */
    MbInfo *a = orc_sl_mb_at(s, s->mb_x - 1, s->mb_y), *b = orc_sl_mb_at(s, s->mb_x, s->mb_y - 1);
    return decision(&s->c, 399 + (a && a->t8x8) + (b && b->t8x8));
}

int orc_cabac_intra_pred_mode(Sl *s) {
    Cabac *c = &s->c;
    if (decision(c, 68)) return -1;                /* prev_intra{4x4,8x8}_pred_mode_flag */
    int v = decision(c, 69);
    v |= decision(c, 69) << 1; v |= decision(c, 69) << 2;
    return v;
}

int orc_cabac_chroma_pred_mode(Sl *s) {           /* 9.3.3.1.1.8, TU cMax 3 */
    Cabac *c = &s->c;
    MbInfo *a = orc_sl_mb_at(s, s->mb_x - 1, s->mb_y), *b = orc_sl_mb_at(s, s->mb_x, s->mb_y - 1);
    int inc = (a && a->is_intra && !a->is_pcm && a->chroma_pred_mode != 0) + (b && b->is_intra && !b->is_pcm && b->chroma_pred_mode != 0);
    if (!decision(c, 64 + inc)) return 0;
    if (!decision(c, 64 + 3)) return 1;
    return 2 + decision(c, 64 + 3);
}

int orc_cabac_ref_idx(Sl *s, int list, int bx, int by) {     /* 9.3.3.1.1.6 */
    Cabac *c = &s->c;
    int inc = 0;
    for (int k = 0; k < 2; k++) {
        int r; MbInfo *m = nb4(s, bx, by, k == 0, &r);
        if (!m || m->is_intra) continue;
        int b8 = (r >> 3) * 2 + ((r & 3) >> 1);
        if (m->direct8 & (1 << b8)) continue;                /* B_Skip / B_Direct / direct sub-macroblock: predicted, not parsed */
        if (m->ref_idx[list][b8] > 0) inc += k == 0 ? 1 : 2;
    }
    int v = 0, ctx = 54 + inc;
    while (decision(c, ctx)) {
        v++;
        ctx = 54 + (v == 1 ? 4 : 5);
        if (v > 32 || c->b->err) return -1;
    }
    return v;
}

int orc_cabac_mvd(Sl *s, int list, int bx, int by, int comp) {   /* 9.3.3.1.1.7, UEG3 uCoff 9 signed */
    Cabac *c = &s->c;
    int sum = 0;
    for (int k = 0; k < 2; k++) {
        int r; MbInfo *m = nb4(s, bx, by, k == 0, &r);
        if (m && !m->is_intra) sum += m->mvd[list][r][comp];     /* predFlagLX == 0 / skip / direct: stored as 0 */
    }
    int base = comp ? 47 : 40;
    int inc = sum < 3 ? 0 : (sum > 32 ? 2 : 1);
    if (!decision(c, base + inc)) return 0;
    int v = 1, ctx = 3;
    while (v < 9 && decision(c, base + ctx)) { v++; if (ctx < 6) ctx++; }
    if (v == 9) {                                               /* Exp-Golomb k = 3 suffix */
        int k = 3;
        while (bypass(c)) { v += 1 << k; k++; if (k > 24 || c->b->err) return 0x7fffffff; }
        while (k--) v += bypass(c) << k;
    }
    return bypass(c) ? -v : v;
}

int orc_cabac_cbp(Sl *s) {                        /* 9.3.3.1.1.4 */
    Cabac *c = &s->c;
    MbInfo *a = orc_sl_mb_at(s, s->mb_x - 1, s->mb_y), *b = orc_sl_mb_at(s, s->mb_x, s->mb_y - 1);
    int cbp = 0;
    for (int b8 = 0; b8 < 4; b8++) {
        int ca, cb;
        if (b8 & 1) ca = !((cbp >> (b8 - 1)) & 1); else ca = a ? !((a->cbp >> (b8 + 1)) & 1) : 0;
        if (b8 & 2) cb = !((cbp >> (b8 - 2)) & 1); else cb = b ? !((b->cbp >> (b8 + 2)) & 1) : 0;
        cbp |= decision(c, 73 + ca + 2 * cb) << b8;
    }
    int ca = a && (a->cbp >> 4) != 0, cb = b && (b->cbp >> 4) != 0;     /* I_PCM carries cbp 0x2f, skip carries 0 */
    if (decision(c, 77 + ca + 2 * cb)) {
        ca = a && (a->cbp >> 4) == 2; cb = b && (b->cbp >> 4) == 2;
        cbp |= (1 + decision(c, 77 + 4 + ca + 2 * cb)) << 4;
    }
    return cbp;
}

int orc_cabac_qp_delta(Sl *s) {                   /* 9.3.3.1.1.5; Table 9-3 mapping */
    Cabac *c = &s->c;
    int ctx = 60 + (s->last_dqp_nonzero ? 1 : 0), k = 0;
    while (decision(c, ctx)) {
        k++;
        ctx = 60 + (k == 1 ? 2 : 3);
        if (k > 104 || c->b->err) return 0x7fffffff;
    }
    return (k & 1) ? (k + 1) >> 1 : -(k >> 1);
}

/* 7.3.5.3.3 residual_block_cabac() */
int orc_cabac_residual_block(Sl *s, int cat, int idx, int16_t *coef, int maxnum) {
    static const int cbf_off[5] = {0, 4, 8, 12, 16}, sig_off[5] = {0, 15, 29, 44, 47}, abs_off[5] = {0, 10, 20, 30, 39};
    Cabac *c = &s->c; MbInfo *mb = s->mb;
    memset(coef, 0, sizeof(int16_t) * maxnum);
    int bit = -1;
    if (cat != 5) {
        /* coded_block_flag, 9.3.3.1.1.9 */
        MbInfo *ma, *mbb; int fa, fb;
        if (cat == 0) {
            bit = 16; ma = orc_sl_mb_at(s, s->mb_x - 1, s->mb_y); mbb = orc_sl_mb_at(s, s->mb_x, s->mb_y - 1);
            fa = ma ? (int)((ma->cbf >> 16) & 1) : -1; fb = mbb ? (int)((mbb->cbf >> 16) & 1) : -1;
        } else if (cat == 1 || cat == 2) {
            int r, bx = idx & 3, by = idx >> 2;
            bit = idx;
            ma = nb4(s, bx, by, 1, &r); fa = ma ? (int)((ma->cbf >> r) & 1) : -1;
            mbb = nb4(s, bx, by, 0, &r); fb = mbb ? (int)((mbb->cbf >> r) & 1) : -1;
        } else if (cat == 3) {
            bit = 17 + idx; ma = orc_sl_mb_at(s, s->mb_x - 1, s->mb_y); mbb = orc_sl_mb_at(s, s->mb_x, s->mb_y - 1);
            fa = ma ? (int)((ma->cbf >> bit) & 1) : -1; fb = mbb ? (int)((mbb->cbf >> bit) & 1) : -1;
        } else {
            int pl = idx >> 2, k = idx & 3, bx = k & 1, by = k >> 1;
            bit = 19 + idx;
            if (bx > 0) { ma = mb; fa = (mb->cbf >> (19 + pl * 4 + by * 2)) & 1; }
            else { ma = orc_sl_mb_at(s, s->mb_x - 1, s->mb_y); fa = ma ? (int)((ma->cbf >> (19 + pl * 4 + by * 2 + 1)) & 1) : -1; }
            if (by > 0) { mbb = mb; fb = (mb->cbf >> (19 + pl * 4 + bx)) & 1; }
            else { mbb = orc_sl_mb_at(s, s->mb_x, s->mb_y - 1); fb = mbb ? (int)((mbb->cbf >> (19 + pl * 4 + 2 + bx)) & 1) : -1; }
        }
        if (fa < 0) fa = mb->is_intra ? 1 : 0;      /* neighbour MB not available */
        if (fb < 0) fb = mb->is_intra ? 1 : 0;      /* (I_PCM neighbours carry all-ones cbf) */
        if (!decision(c, 85 + cbf_off[cat] + fa + 2 * fb)) return 0;
        mb->cbf |= 1u << bit;
    }
    /* Table 9-34: significant_coeff_flag / last_significant_coeff_flag of field-coded blocks have their own contexts, 277.. / 338.. (the
     * 8x8 blocks' 436.. / 451.. are not in the tables: their initial values are not pinned, tests/SPEC_AUDIT.md, and such streams are refused) */
    const int fld = s->d->field_pic;
    if (fld && cat == 5) { snprintf(s->d->err, sizeof s->d->err, "CABAC residual of a field-coded 8x8 block unsupported"); return -1; }
    int sig_base = cat == 5 ? 402 : (fld ? 277 : 105) + sig_off[cat], last_base = cat == 5 ? 417 : (fld ? 338 : 166) + sig_off[cat];
    int abs_base = cat == 5 ? 426 : 227 + abs_off[cat];
    uint8_t sig[64];
    int num = maxnum, i;
    memset(sig, 0, sizeof sig);
    for (i = 0; i < num - 1; i++) {
        int si = cat == 5 ? orc_cabac_sig8_inc[i] : (cat == 3 ? orc_min(i, 2) : i);
        if (decision(c, sig_base + si)) {
            sig[i] = 1;
            int li = cat == 5 ? orc_cabac_last8_inc[i] : (cat == 3 ? orc_min(i, 2) : i);
            if (decision(c, last_base + li)) { num = i + 1; break; }
        }
        if (c->b->err) return -1;
    }
    if (i == maxnum - 1) sig[maxnum - 1] = 1;       /* last coefficient inferred significant */
    int n = 0, gt1 = 0, eq1 = 0;
    for (i = num - 1; i >= 0; i--) {
        if (!sig[i]) continue;
        int ctx = abs_base + (gt1 ? 0 : orc_min(4, 1 + eq1)), v = 0;
        if (decision(c, ctx)) {
            v = 1;
            ctx = abs_base + 5 + orc_min(4 - (cat == 3), gt1);
            while (v < 14 && decision(c, ctx)) v++;
            if (v == 14) {                          /* UEG0 suffix */
                int k = 0;
                while (bypass(c)) { v += 1 << k; k++; if (k > 20 || c->b->err) return -1; }
                while (k--) v += bypass(c) << k;
            }
        }
        int level = v + 1;
        if (level == 1) eq1++; else gt1++;
        if (level > 32768) return -1;
        coef[i] = (int16_t)(bypass(c) ? -level : level);
        n++;
    }
    return c->b->err ? -1 : n;
}
