/*
 * oracle/orc_slice.c -- CPU ORACLE (test infrastructure only).
 * Slice data: CAVLC macroblock parsing (7.3.4, 7.3.5, 9.2), intra prediction
 * (8.3), inter prediction (8.4) and the transform / reconstruction path (8.5).
 * This is the restatement of the closed picture reconstruction performed by
 * cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:37) for progressive
 * 8-bit 4:2:0 I/P slices.
 */
#include "orc_slice.h"
#include "orc_tables.h"

/* ------------------------------------------------------------------------ */
MbInfo *orc_sl_mb_at(Sl *s, int mx, int my) {
    if (mx < 0 || my < 0 || mx >= s->d->mb_w || my >= s->d->mb_h) return NULL;
    MbInfo *m = &s->pic->mbs[my * s->d->mb_w + mx];
    return m->slice_num == s->d->slice_num ? m : NULL;     /* 6.4.x availability */
}
#define mb_at orc_sl_mb_at
static int intra_usable(Sl *s, MbInfo *m) {               /* for intra prediction */
    if (!m) return 0;
    if (s->pps->constrained_intra_pred && !m->is_intra) return 0;
    return 1;
}

/* ------------------------------- CAVLC ---------------------------------- */
static int vlc_match(Bits *b, const uint8_t *len, const uint8_t *bits, int n) {
    /* read one bit at a time until the prefix equals exactly one codeword */
    unsigned v = 0;
    for (int l = 1; l <= 16; l++) {
        v = (v << 1) | bits_u1(b);
        if (b->err) return -1;
        for (int i = 0; i < n; i++) if (len[i] == l && bits[i] == v) return i;
    }
    return -1;
}

/* 9.2.1: returns total_coeff; fills coef[0..max_num-1] in scan order */
static int residual_block_cavlc(Sl *s, int nC, int max_num, int16_t *coef) {
    Bits *b = s->b;
    int idx;
    memset(coef, 0, sizeof(int16_t) * max_num);
    if (nC == -1) idx = vlc_match(b, orc_chroma_dc_token_len, orc_chroma_dc_token_bits, 20);
    else {
        int t = nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;
        idx = vlc_match(b, orc_coeff_token_len[t], orc_coeff_token_bits[t], 68);
    }
    if (idx < 0) return -1;
    int total = idx >> 2, t1 = idx & 3;
    if (total == 0) return 0;
    if (total > max_num) return -1;
    int level[16], run[16];
    int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
    for (int i = 0; i < total; i++) {
        if (i < t1) level[i] = 1 - 2 * (int)bits_u1(b);
        else {
            int prefix = 0;
            while (!bits_u1(b)) { if (b->err || ++prefix > 32) return -1; }
            int code = orc_min(15, prefix) << suffix_len;
            if (suffix_len > 0 || prefix >= 14) {
                int size = (prefix == 14 && suffix_len == 0) ? 4 : (prefix >= 15 ? prefix - 3 : suffix_len);
                if (size > 0) code += (int)bits_u(b, size);
            }
            if (prefix >= 15 && suffix_len == 0) code += 15;
            if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
            if (i == t1 && t1 < 3) code += 2;
            level[i] = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
            if (suffix_len == 0) suffix_len = 1;
            if (orc_abs(level[i]) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
        }
    }
    int zeros_left = 0;
    if (total < max_num) {
        int tz;
        if (max_num == 4) tz = vlc_match(b, orc_cdc_total_zeros_len[total - 1], orc_cdc_total_zeros_bits[total - 1], 4 - total + 1);
        else tz = vlc_match(b, orc_total_zeros_len[total - 1], orc_total_zeros_bits[total - 1], 16 - total + 1);
        if (tz < 0) return -1;
        zeros_left = tz;
    }
    for (int i = 0; i < total - 1; i++) {
        if (zeros_left > 0) {
            int t = orc_min(zeros_left, 7) - 1;
            int r = vlc_match(b, orc_run_len[t], orc_run_bits[t], orc_run_count[t]);
            if (r < 0 || r > zeros_left) return -1;
            run[i] = r; zeros_left -= r;
        } else run[i] = 0;
    }
    run[total - 1] = zeros_left;
    int pos = -1;
    for (int i = total - 1; i >= 0; i--) {
        pos += run[i] + 1;
        if (pos >= max_num) return -1;
        coef[pos] = (int16_t)level[i];
    }
    return b->err ? -1 : total;
}

/* 9.2.1 nC for a luma 4x4 block at raster (bx,by) / chroma block (plane 1,2) */
static int nc_luma(Sl *s, int bx, int by) {
    int availA = 0, availB = 0, nA = 0, nB = 0;
    if (bx > 0) { availA = 1; nA = s->mb->total_coeff[by * 4 + bx - 1]; }
    else { MbInfo *m = mb_at(s, s->mb_x - 1, s->mb_y); if (m) { availA = 1; nA = m->total_coeff[by * 4 + 3]; } }
    if (by > 0) { availB = 1; nB = s->mb->total_coeff[(by - 1) * 4 + bx]; }
    else { MbInfo *m = mb_at(s, s->mb_x, s->mb_y - 1); if (m) { availB = 1; nB = m->total_coeff[12 + bx]; } }
    if (availA && availB) return (nA + nB + 1) >> 1;
    return availA ? nA : (availB ? nB : 0);
}
static int nc_chroma(Sl *s, int pl, int bx, int by) {
    int o = 16 + 4 * pl, availA = 0, availB = 0, nA = 0, nB = 0;
    if (bx > 0) { availA = 1; nA = s->mb->total_coeff[o + by * 2 + bx - 1]; }
    else { MbInfo *m = mb_at(s, s->mb_x - 1, s->mb_y); if (m) { availA = 1; nA = m->total_coeff[o + by * 2 + 1]; } }
    if (by > 0) { availB = 1; nB = s->mb->total_coeff[o + (by - 1) * 2 + bx]; }
    else { MbInfo *m = mb_at(s, s->mb_x, s->mb_y - 1); if (m) { availB = 1; nB = m->total_coeff[o + 2 + bx]; } }
    if (availA && availB) return (nA + nB + 1) >> 1;
    return availA ? nA : (availB ? nB : 0);
}

static inline int blk_x(int blk) { return (blk & 1) + 2 * ((blk >> 2) & 1); }
static inline int blk_y(int blk) { return ((blk >> 1) & 1) + 2 * (blk >> 3); }

/* 7.3.5.3 residual() with CAVLC */
static int parse_residual(Sl *s, int cbp) {
    MbInfo *mb = s->mb;
    int16_t tmp[16];
    /* 8.5.6 / 8.5.7: the macroblocks of a field picture are field macroblocks: field scan */
    const uint8_t *z4 = s->d->field_pic ? orc_fieldscan4 : orc_zigzag4, *z8 = s->d->field_pic ? orc_fieldscan8 : orc_zigzag8;
    if (mb->is_i16) {
        int n = residual_block_cavlc(s, nc_luma(s, 0, 0), 16, tmp);
        if (n < 0) return -1;
        for (int i = 0; i < 16; i++) s->i16dc[z4[i]] = tmp[i];
    }
    for (int b8 = 0; b8 < 4; b8++) {
        for (int k = 0; k < 4; k++) {
            int blk = b8 * 4 + k, bx = blk_x(blk), by = blk_y(blk), r = by * 4 + bx;
            if (!(cbp & (1 << b8))) { mb->total_coeff[r] = 0; continue; }
            int nC = nc_luma(s, bx, by), n;
            if (mb->is_i16) {
                n = residual_block_cavlc(s, nC, 15, tmp);
                if (n < 0) return -1;
                for (int i = 0; i < 15; i++) s->luma[r][z4[i + 1]] = tmp[i];
            } else {
                n = residual_block_cavlc(s, nC, 16, tmp);
                if (n < 0) return -1;
                if (mb->t8x8) {   /* 7.3.5.3.2: 4x4 block k carries 8x8 scan positions 4*i+k */
                    for (int i = 0; i < 16; i++) s->luma8[b8][z8[4 * i + k]] = tmp[i];
                } else
                    for (int i = 0; i < 16; i++) s->luma[r][z4[i]] = tmp[i];
            }
            mb->total_coeff[r] = (uint8_t)n;
        }
    }
    if (cbp & 0x30) {
        for (int pl = 0; pl < 2; pl++)
            if (residual_block_cavlc(s, -1, 4, s->cdc[pl]) < 0) return -1;
    }
    for (int pl = 0; pl < 2; pl++)
        for (int k = 0; k < 4; k++) {
            if (!(cbp & 0x20)) { mb->total_coeff[16 + 4 * pl + k] = 0; continue; }
            int n = residual_block_cavlc(s, nc_chroma(s, pl, k & 1, k >> 1), 15, tmp);
            if (n < 0) return -1;
            for (int i = 0; i < 15; i++) s->cac[pl][k][z4[i + 1]] = tmp[i];
            mb->total_coeff[16 + 4 * pl + k] = (uint8_t)n;
        }
    return 0;
}

/* ----------------------------- transforms -------------------------------- */
static int chroma_qp(const Pps *pps, int qpy, int pl) {
    int off = pl == 0 ? pps->chroma_qp_index_offset : pps->second_chroma_qp_index_offset;
    int qpi = orc_clip3(0, 51, qpy + off);
    return qpi < 30 ? qpi : orc_qpc_tab[qpi - 30];
}
/* LevelScale4x4(m,i,j) = weightScale4x4 * normAdjust4x4 (8.5.9); list in zig-zag order */
static int level_scale4(const uint8_t *list_zz, int m, int raster) {
    int w = 16;
    for (int k = 0; k < 16; k++) if (orc_zigzag4[k] == raster) { w = list_zz[k]; break; }
    int i = raster >> 2, j = raster & 3;
    int cls = (!(i & 1) && !(j & 1)) ? 0 : ((i & 1) && (j & 1)) ? 1 : 2;
    return w * orc_norm4[m][cls];
}
static int level_scale8(const uint8_t *list_zz, int m, int raster) {
    int w = 16;
    for (int k = 0; k < 64; k++) if (orc_zigzag8[k] == raster) { w = list_zz[k]; break; }
    int i = raster >> 3, j = raster & 7, cls;
    if (i % 4 == 0 && j % 4 == 0) cls = 0;
    else if (i % 2 == 1 && j % 2 == 1) cls = 1;
    else if (i % 4 == 2 && j % 4 == 2) cls = 2;
    else if ((i % 4 == 0 && j % 2 == 1) || (i % 2 == 1 && j % 4 == 0)) cls = 3;
    else if ((i % 4 == 0 && j % 4 == 2) || (i % 4 == 2 && j % 4 == 0)) cls = 4;
    else cls = 5;
    return w * orc_norm8[m][cls];
}

/* 8.5.12.1 scaling of a 4x4 residual block; dc_done: element 0 already scaled */
static void scale4x4(const int16_t *c, int *dq, const uint8_t *list, int qp, int dc_done) {
    for (int k = 0; k < 16; k++) {
        if (k == 0 && dc_done) { dq[0] = c[0]; continue; }
        int ls = level_scale4(list, qp % 6, k);
        if (qp >= 24) dq[k] = (c[k] * ls) << (qp / 6 - 4);
        else dq[k] = (c[k] * ls + (1 << (3 - qp / 6))) >> (4 - qp / 6);
    }
}
/* 8.5.12.2 inverse 4x4 transform -> residual r (before the >>6 already applied) */
static void idct4x4(const int *dq, int *r) {
    int f[16];
    for (int i = 0; i < 4; i++) {                 /* rows */
        const int *d = dq + 4 * i;
        int e0 = d[0] + d[2], e1 = d[0] - d[2], e2 = (d[1] >> 1) - d[3], e3 = d[1] + (d[3] >> 1);
        f[4 * i + 0] = e0 + e3; f[4 * i + 1] = e1 + e2; f[4 * i + 2] = e1 - e2; f[4 * i + 3] = e0 - e3;
    }
    for (int j = 0; j < 4; j++) {                 /* columns */
        int g0 = f[j] + f[8 + j], g1 = f[j] - f[8 + j], g2 = (f[4 + j] >> 1) - f[12 + j], g3 = f[4 + j] + (f[12 + j] >> 1);
        r[j] = (g0 + g3 + 32) >> 6; r[4 + j] = (g1 + g2 + 32) >> 6;
        r[8 + j] = (g1 - g2 + 32) >> 6; r[12 + j] = (g0 - g3 + 32) >> 6;
    }
}
static void idct8_1d(const int *d, int *o, int stride_in, int stride_out) {
    int d0 = d[0], d1 = d[stride_in], d2 = d[2 * stride_in], d3 = d[3 * stride_in];
    int d4 = d[4 * stride_in], d5 = d[5 * stride_in], d6 = d[6 * stride_in], d7 = d[7 * stride_in];
    int e0 = d0 + d4, e1 = -d3 + d5 - d7 - (d7 >> 1), e2 = d0 - d4, e3 = d1 + d7 - d3 - (d3 >> 1);
    int e4 = (d2 >> 1) - d6, e5 = -d1 + d7 + d5 + (d5 >> 1), e6 = d2 + (d6 >> 1), e7 = d3 + d5 + d1 + (d1 >> 1);
    int f0 = e0 + e6, f1 = e1 + (e7 >> 2), f2 = e2 + e4, f3 = e3 + (e5 >> 2);
    int f4 = e2 - e4, f5 = (e3 >> 2) - e5, f6 = e0 - e6, f7 = e7 - (e1 >> 2);
    o[0] = f0 + f7; o[stride_out] = f2 + f5; o[2 * stride_out] = f4 + f3; o[3 * stride_out] = f6 + f1;
    o[4 * stride_out] = f6 - f1; o[5 * stride_out] = f4 - f3; o[6 * stride_out] = f2 - f5; o[7 * stride_out] = f0 - f7;
}
static void idct8x8(const int *dq, int *r) {
    int t[64];
    for (int i = 0; i < 8; i++) idct8_1d(dq + 8 * i, t + 8 * i, 1, 1);
    for (int j = 0; j < 8; j++) idct8_1d(t + j, r + j, 8, 8);
    for (int k = 0; k < 64; k++) r[k] = (r[k] + 32) >> 6;
}
static void add_block(uint8_t *dst, int stride, const int *r, int n) {
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) dst[y * stride + x] = (uint8_t)orc_clip1(dst[y * stride + x] + r[y * n + x]);
}

/* ----------------------------- intra prediction -------------------------- */
/* gather neighbours of a 4x4 block at pixel (px,py) of plane; T[-1]=corner */
static void pred4x4(Sl *s, int blk, int mode, uint8_t *dst, int stride) {
    int bx = blk_x(blk), by = blk_y(blk);
    MbInfo *mA = bx > 0 ? s->mb : mb_at(s, s->mb_x - 1, s->mb_y);
    MbInfo *mB = by > 0 ? s->mb : mb_at(s, s->mb_x, s->mb_y - 1);
    MbInfo *mD = (bx > 0 && by > 0) ? s->mb : (bx > 0 ? mb_at(s, s->mb_x, s->mb_y - 1)
                 : (by > 0 ? mb_at(s, s->mb_x - 1, s->mb_y) : mb_at(s, s->mb_x - 1, s->mb_y - 1)));
    int availA = bx > 0 || intra_usable(s, mA), availB = by > 0 || intra_usable(s, mB);
    int availD = (bx > 0 && by > 0) || intra_usable(s, mD);
    int availC;
    if (by == 0) {
        MbInfo *mC = bx < 3 ? mb_at(s, s->mb_x, s->mb_y - 1) : mb_at(s, s->mb_x + 1, s->mb_y - 1);
        availC = intra_usable(s, mC);
    } else availC = !(bx == 3 || blk == 3 || blk == 11 || blk == 7 || blk == 13 || blk == 15);
    int Tbuf[9], Lbuf[5], *T = Tbuf + 1, *L = Lbuf + 1;
    for (int i = 0; i < 4; i++) { T[i] = availB ? dst[-stride + i] : 128; L[i] = availA ? dst[i * stride - 1] : 128; }
    for (int i = 4; i < 8; i++) T[i] = (availB && availC) ? dst[-stride + i] : T[3];
    T[-1] = L[-1] = availD ? dst[-stride - 1] : 128;
    int p[16];
    switch (mode) {
    case 0: for (int k = 0; k < 16; k++) p[k] = T[k & 3]; break;
    case 1: for (int k = 0; k < 16; k++) p[k] = L[k >> 2]; break;
    case 2: {
        int dc;
        if (availA && availB) dc = (T[0] + T[1] + T[2] + T[3] + L[0] + L[1] + L[2] + L[3] + 4) >> 3;
        else if (availA) dc = (L[0] + L[1] + L[2] + L[3] + 2) >> 2;
        else if (availB) dc = (T[0] + T[1] + T[2] + T[3] + 2) >> 2;
        else dc = 128;
        for (int k = 0; k < 16; k++) p[k] = dc;
        break; }
    case 3: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++)
                p[y * 4 + x] = (x == 3 && y == 3) ? (T[6] + 3 * T[7] + 2) >> 2
                                                  : (T[x + y] + 2 * T[x + y + 1] + T[x + y + 2] + 2) >> 2;
            break;
    case 4: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
                if (x > y) p[y * 4 + x] = (T[x - y - 2] + 2 * T[x - y - 1] + T[x - y] + 2) >> 2;
                else if (x < y) p[y * 4 + x] = (L[y - x - 2] + 2 * L[y - x - 1] + L[y - x] + 2) >> 2;
                else p[y * 4 + x] = (T[0] + 2 * T[-1] + L[0] + 2) >> 2;
            } break;
    case 5: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
                int z = 2 * x - y, i = x - (y >> 1);
                if (z >= 0 && !(z & 1)) p[y * 4 + x] = (T[i - 1] + T[i] + 1) >> 1;
                else if (z >= 0) p[y * 4 + x] = (T[i - 2] + 2 * T[i - 1] + T[i] + 2) >> 2;
                else if (z == -1) p[y * 4 + x] = (L[0] + 2 * T[-1] + T[0] + 2) >> 2;
                else p[y * 4 + x] = (L[y - 1] + 2 * L[y - 2] + L[y - 3] + 2) >> 2;
            } break;
    case 6: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
                int z = 2 * y - x, i = y - (x >> 1);
                if (z >= 0 && !(z & 1)) p[y * 4 + x] = (L[i - 1] + L[i] + 1) >> 1;
                else if (z >= 0) p[y * 4 + x] = (L[i - 2] + 2 * L[i - 1] + L[i] + 2) >> 2;
                else if (z == -1) p[y * 4 + x] = (L[0] + 2 * T[-1] + T[0] + 2) >> 2;
                else p[y * 4 + x] = (T[x - 1] + 2 * T[x - 2] + T[x - 3] + 2) >> 2;
            } break;
    case 7: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
                int i = x + (y >> 1);
                p[y * 4 + x] = (y & 1) ? (T[i] + 2 * T[i + 1] + T[i + 2] + 2) >> 2 : (T[i] + T[i + 1] + 1) >> 1;
            } break;
    default: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
                int z = x + 2 * y, i = y + (x >> 1);
                if (z > 5) p[y * 4 + x] = L[3];
                else if (z == 5) p[y * 4 + x] = (L[2] + 3 * L[3] + 2) >> 2;
                else if (z & 1) p[y * 4 + x] = (L[i] + 2 * L[i + 1] + L[i + 2] + 2) >> 2;
                else p[y * 4 + x] = (L[i] + L[i + 1] + 1) >> 1;
            } break;
    }
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) dst[y * stride + x] = (uint8_t)p[y * 4 + x];
}

/* 8.3.3 Intra16x16 / 8.3.4 chroma (n = 16 or 8) */
static void pred_plane_block(uint8_t *dst, int stride, int n) {
    int H = 0, V = 0, half = n / 2;
    for (int k = 0; k < half; k++) {
        int t0 = dst[-stride + half + k], t1 = (half - 2 - k) >= 0 ? dst[-stride + half - 2 - k] : dst[-stride - 1];
        int l0 = dst[(half + k) * stride - 1], l1 = (half - 2 - k) >= 0 ? dst[(half - 2 - k) * stride - 1] : dst[-stride - 1];
        H += (k + 1) * (t0 - t1); V += (k + 1) * (l0 - l1);
    }
    int a = 16 * (dst[(n - 1) * stride - 1] + dst[-stride + n - 1]);
    int bb = n == 16 ? (5 * H + 32) >> 6 : (34 * H + 32) >> 6;
    int cc = n == 16 ? (5 * V + 32) >> 6 : (34 * V + 32) >> 6;
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++)
        dst[y * stride + x] = (uint8_t)orc_clip1((a + bb * (x - (half - 1)) + cc * (y - (half - 1)) + 16) >> 5);
}
static int pred16x16(Sl *s, int mode, uint8_t *dst, int stride) {
    int availA = intra_usable(s, mb_at(s, s->mb_x - 1, s->mb_y));
    int availB = intra_usable(s, mb_at(s, s->mb_x, s->mb_y - 1));
    int availD = intra_usable(s, mb_at(s, s->mb_x - 1, s->mb_y - 1));
    switch (mode) {
    case 0: if (!availB) return -1;
        for (int y = 0; y < 16; y++) memcpy(dst + y * stride, dst - stride, 16); break;
    case 1: if (!availA) return -1;
        for (int y = 0; y < 16; y++) memset(dst + y * stride, dst[y * stride - 1], 16); break;
    case 2: {
        int st = 0, sl = 0, dc;
        for (int i = 0; i < 16; i++) { if (availB) st += dst[-stride + i]; if (availA) sl += dst[i * stride - 1]; }
        if (availA && availB) dc = (st + sl + 16) >> 5;
        else if (availA) dc = (sl + 8) >> 4; else if (availB) dc = (st + 8) >> 4; else dc = 128;
        for (int y = 0; y < 16; y++) memset(dst + y * stride, dc, 16);
        break; }
    default: if (!availA || !availB || !availD) return -1;
        pred_plane_block(dst, stride, 16); break;
    }
    return 0;
}
static int pred_chroma(Sl *s, int mode, uint8_t *dst, int stride) {
    int availA = intra_usable(s, mb_at(s, s->mb_x - 1, s->mb_y));
    int availB = intra_usable(s, mb_at(s, s->mb_x, s->mb_y - 1));
    int availD = intra_usable(s, mb_at(s, s->mb_x - 1, s->mb_y - 1));
    switch (mode) {
    case 0:
        for (int by = 0; by < 2; by++) for (int bx = 0; bx < 2; bx++) {
            int st = 0, sl = 0, dc;
            for (int i = 0; i < 4; i++) { if (availB) st += dst[-stride + bx * 4 + i]; if (availA) sl += dst[(by * 4 + i) * stride - 1]; }
            if (bx == by) {                       /* (0,0) and (1,1) */
                if (availA && availB) dc = (st + sl + 4) >> 3;
                else if (availA) dc = (sl + 2) >> 2; else if (availB) dc = (st + 2) >> 2; else dc = 128;
            } else if (bx == 1) {                 /* (1,0): prefer top */
                if (availB) dc = (st + 2) >> 2; else if (availA) dc = (sl + 2) >> 2; else dc = 128;
            } else {                              /* (0,1): prefer left */
                if (availA) dc = (sl + 2) >> 2; else if (availB) dc = (st + 2) >> 2; else dc = 128;
            }
            for (int y = 0; y < 4; y++) memset(dst + (by * 4 + y) * stride + bx * 4, dc, 4);
        }
        break;
    case 1: if (!availA) return -1;
        for (int y = 0; y < 8; y++) memset(dst + y * stride, dst[y * stride - 1], 8); break;
    case 2: if (!availB) return -1;
        for (int y = 0; y < 8; y++) memcpy(dst + y * stride, dst - stride, 8); break;
    default: if (!availA || !availB || !availD) return -1;
        pred_plane_block(dst, stride, 8); break;
    }
    return 0;
}

/* ----------------------------- inter prediction -------------------------- */
typedef struct { int avail; int ref; int mv[2]; } Nb;

/* neighbour partition covering 4x4 block (bx,by) relative to the current MB */
static Nb nb_get(Sl *s, int list, int bx, int by) {
    Nb n = {0, -1, {0, 0}};
    MbInfo *m; int rx, ry;
    if (by < 0) {
        if (bx < 0) { m = mb_at(s, s->mb_x - 1, s->mb_y - 1); rx = 3; }
        else if (bx > 3) { m = mb_at(s, s->mb_x + 1, s->mb_y - 1); rx = bx - 4; }
        else { m = mb_at(s, s->mb_x, s->mb_y - 1); rx = bx; }
        ry = 3;
    } else if (bx < 0) { m = mb_at(s, s->mb_x - 1, s->mb_y); rx = 3; ry = by; }
    else if (bx > 3) return n;
    else {
        if (!(s->decoded_mask & (1 << (by * 4 + bx)))) return n;
        m = s->mb; rx = bx; ry = by;
    }
    if (!m) return n;
    n.avail = 1;
    if (m->is_intra) return n;
    n.ref = m->ref_idx[list][(ry >> 1) * 2 + (rx >> 1)];
    if (n.ref >= 0) { n.mv[0] = m->mv[list][ry * 4 + rx][0]; n.mv[1] = m->mv[list][ry * 4 + rx][1]; }
    return n;
}
static int median3(int a, int b, int c) {
    int mx = orc_max(a, orc_max(b, c)), mn = orc_min(a, orc_min(b, c));
    return a + b + c - mx - mn;
}
/* 8.4.1.3: mv prediction for a partition at 4x4 (bx,by) size (bw,bh) in 4x4 units */
static void predict_mv(Sl *s, int list, int bx, int by, int bw, int bh, int ref, int shape, int part, int mvp[2]) {
    Nb A = nb_get(s, list, bx - 1, by), B = nb_get(s, list, bx, by - 1), C = nb_get(s, list, bx + bw, by - 1);
    if (!C.avail) C = nb_get(s, list, bx - 1, by - 1);
    (void)bh;
    if (shape == 1) {            /* 16x8 */
        if (part == 0 && B.ref == ref) { mvp[0] = B.mv[0]; mvp[1] = B.mv[1]; return; }
        if (part == 1 && A.ref == ref) { mvp[0] = A.mv[0]; mvp[1] = A.mv[1]; return; }
    } else if (shape == 2) {     /* 8x16 */
        if (part == 0 && A.ref == ref) { mvp[0] = A.mv[0]; mvp[1] = A.mv[1]; return; }
        if (part == 1 && C.ref == ref) { mvp[0] = C.mv[0]; mvp[1] = C.mv[1]; return; }
    }
    if (!B.avail && !C.avail && A.avail) { B = A; C = A; }
    int ma = A.ref == ref, mb = B.ref == ref, mc = C.ref == ref;
    if (ma + mb + mc == 1) {
        const Nb *n = ma ? &A : (mb ? &B : &C);
        mvp[0] = n->mv[0]; mvp[1] = n->mv[1];
    } else {
        mvp[0] = median3(A.mv[0], B.mv[0], C.mv[0]);
        mvp[1] = median3(A.mv[1], B.mv[1], C.mv[1]);
    }
}
static void set_mv(Sl *s, int list, int bx, int by, int bw, int bh, int mvx, int mvy) {
    for (int y = by; y < by + bh; y++) for (int x = bx; x < bx + bw; x++) {
        s->mb->mv[list][y * 4 + x][0] = (int16_t)mvx; s->mb->mv[list][y * 4 + x][1] = (int16_t)mvy;
        s->decoded_mask |= 1 << (y * 4 + x);
    }
}

static inline int ref_px(const uint8_t *pl, int stride, int w, int h, int x, int y) {
    return pl[orc_clip3(0, h - 1, y) * stride + orc_clip3(0, w - 1, x)];
}
static inline int tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }
/* 8.4.2.2.1 luma sample interpolation for one sample */
static int luma_sample(const Picture *r, int W, int H, int xi, int yi, int fx, int fy) {
    const uint8_t *p = r->y; int st = r->stride_y;
#define P(dx, dy) ref_px(p, st, W, H, xi + (dx), yi + (dy))
#define HB1(dy) tap6(P(-2, dy), P(-1, dy), P(0, dy), P(1, dy), P(2, dy), P(3, dy))   /* b1 at row dy */
#define VH1(dx) tap6(P(dx, -2), P(dx, -1), P(dx, 0), P(dx, 1), P(dx, 2), P(dx, 3))   /* h1 at col dx */
    int G = P(0, 0);
    if (!fx && !fy) return G;
    int b = orc_clip1((HB1(0) + 16) >> 5), h = orc_clip1((VH1(0) + 16) >> 5);
    if (fy == 0) return fx == 2 ? b : (fx == 1 ? (G + b + 1) >> 1 : (P(1, 0) + b + 1) >> 1);
    if (fx == 0) return fy == 2 ? h : (fy == 1 ? (G + h + 1) >> 1 : (P(0, 1) + h + 1) >> 1);
    int j1 = tap6(HB1(-2), HB1(-1), HB1(0), HB1(1), HB1(2), HB1(3));
    int j = orc_clip1((j1 + 512) >> 10);
    int s_ = orc_clip1((HB1(1) + 16) >> 5), m = orc_clip1((VH1(1) + 16) >> 5);
    if (fx == 2 && fy == 2) return j;
    if (fx == 2) return fy == 1 ? (b + j + 1) >> 1 : (j + s_ + 1) >> 1;          /* f, q */
    if (fy == 2) return fx == 1 ? (h + j + 1) >> 1 : (j + m + 1) >> 1;           /* i, k */
    if (fx == 1 && fy == 1) return (b + h + 1) >> 1;                             /* e */
    if (fx == 3 && fy == 1) return (b + m + 1) >> 1;                             /* g */
    if (fx == 1 && fy == 3) return (h + s_ + 1) >> 1;                            /* p */
    return (m + s_ + 1) >> 1;                                                    /* r */
#undef P
#undef HB1
#undef VH1
}
/* 8.4.2.2: prediction samples of one 4x4 block (luma 4x4 + chroma 2x2 per plane) from one reference picture */
static void pred_block(Sl *s, const Picture *ref, int px, int py, int mvx, int mvy, int *yl, int *cu, int *cv) {
    OrcDec *d = s->d;
    int W = d->mb_w * 16, H = d->mb_h * 16;
    int x0 = s->mb_x * 16 + px, y0 = s->mb_y * 16 + py;
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++)
        yl[y * 4 + x] = luma_sample(ref, W, H, x0 + x + (mvx >> 2), y0 + y + (mvy >> 2), mvx & 3, mvy & 3);
    int cw = W / 2, ch = H / 2, cx0 = x0 / 2, cy0 = y0 / 2;
    /* 8.4.1.4, Table 8-9: in a field the chroma vector's vertical component is the luma one, minus 2 (units of 1/8 chroma sample) when a top field
     * predicts from a bottom field, plus 2 when a bottom field predicts from a top field */
    if (d->field_pic && ref->is_field && ref->parity != d->cur_parity) { mvy += d->cur_parity ? 2 : -2; d->stats[ORC_ST_CROSS_PARITY]++; }
    for (int pl = 0; pl < 2; pl++) {
        const uint8_t *rp = pl ? ref->v : ref->u; int *o = pl ? cv : cu;
        for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) {
            int xi = cx0 + x + (mvx >> 3), yi = cy0 + y + (mvy >> 3), fx = mvx & 7, fy = mvy & 7;
            int A = ref_px(rp, ref->stride_c, cw, ch, xi, yi), B = ref_px(rp, ref->stride_c, cw, ch, xi + 1, yi);
            int C = ref_px(rp, ref->stride_c, cw, ch, xi, yi + 1), D = ref_px(rp, ref->stride_c, cw, ch, xi + 1, yi + 1);
            o[y * 2 + x] = ((8 - fx) * (8 - fy) * A + fx * (8 - fy) * B + (8 - fx) * fy * C + fx * fy * D + 32) >> 6;
        }
    }
}
/* 8.4.2.3: weighted sample prediction of n samples; p1 == NULL: one list */
static void weight_samples(int *out, const int *p0, const int *p1, int n, int mode /*0 default 1 explicit 2 implicit*/,
                           int logwd, int w0, int w1, int o0, int o1) {
    for (int i = 0; i < n; i++) {
        int v;
        if (!p1) v = mode == 1 ? orc_clip1((logwd >= 1 ? ((p0[i] * w0 + (1 << (logwd - 1))) >> logwd) : p0[i] * w0) + o0) : p0[i];
        else if (mode == 0) v = (p0[i] + p1[i] + 1) >> 1;
        else v = orc_clip1(((p0[i] * w0 + p1[i] * w1 + (1 << logwd)) >> (logwd + 1)) + ((o0 + o1 + 1) >> 1));
        out[i] = v;
    }
}
/* 8.4.2.3.1 implicit bi-prediction weights of the pair (ref0, ref1) */
static void implicit_weights(const OrcDec *d, const Picture *r0, const Picture *r1, int *w0, int *w1) {
    int tb = orc_clip3(-128, 127, d->cur->poc - r0->poc), td = orc_clip3(-128, 127, r1->poc - r0->poc);
    *w0 = *w1 = 32;
    if (td == 0 || r0->is_ref == 2 || r1->is_ref == 2) return;
    int tx = (16384 + orc_abs(td / 2)) / td, dsf = orc_clip3(-1024, 1023, (tb * tx + 32) >> 6);
    if ((dsf >> 2) < -64 || (dsf >> 2) > 128) return;
    *w0 = 64 - (dsf >> 2); *w1 = dsf >> 2;
}
static int inter_recon(Sl *s) {
    MbInfo *mb = s->mb; OrcDec *d = s->d; Picture *pic = s->pic; const SliceHdr *sh = s->sh;
    int mode = sh->slice_type == SLICE_P ? (s->pps->weighted_pred ? 1 : 0) : s->pps->weighted_bipred_idc;
    /* walk the 4x4 grid, merging nothing: per-4x4 MC gives identical samples */
    for (int b8 = 0; b8 < 4; b8++) {
        const Picture *ref[2] = {NULL, NULL}; int ri[2];
        for (int l = 0; l < 2; l++) {
            ri[l] = mb->ref_idx[l][b8];
            if (ri[l] < 0) continue;
            if (ri[l] >= d->ref_count[l] || !d->ref_list[l][ri[l]]) { snprintf(d->err, sizeof d->err, "missing reference picture (list %d ref_idx %d)", l,
                ri[l]); return -1; }
            ref[l] = d->ref_list[l][ri[l]];
            mb->ref_pic_id[l][b8] = ref[l]->id;
        }
        if (!ref[0] && !ref[1]) { snprintf(d->err, sizeof d->err, "inter block without a reference"); return -1; }
        for (int k = 0; k < 4; k++) {
            int bx = (b8 & 1) * 2 + (k & 1), by = (b8 >> 1) * 2 + (k >> 1), r = by * 4 + bx;
            int y[2][16], cu[2][4], cv[2][4], oy[16], ou[4], ov[4];
            for (int l = 0; l < 2; l++) if (ref[l]) pred_block(s, ref[l], bx * 4, by * 4, mb->mv[l][r][0], mb->mv[l][r][1], y[l], cu[l], cv[l]);
            if (ref[0] && ref[1]) {
                int m = mode, w0 = 32, w1 = 32;
                if (m == 2) { implicit_weights(d, ref[0], ref[1], &w0, &w1); }
                if (m == 1) {
                    weight_samples(oy, y[0], y[1], 16, 1, sh->luma_log2_wd, sh->luma_weight[0][ri[0]], sh->luma_weight[1][ri[1]], sh->luma_offset[0][ri[0]],
                        sh->luma_offset[1][ri[1]]);
                    weight_samples(ou, cu[0], cu[1], 4, 1, sh->chroma_log2_wd, sh->chroma_weight[0][ri[0]][0], sh->chroma_weight[1][ri[1]][0],
                        sh->chroma_offset[0][ri[0]][0], sh->chroma_offset[1][ri[1]][0]);
                    weight_samples(ov, cv[0], cv[1], 4, 1, sh->chroma_log2_wd, sh->chroma_weight[0][ri[0]][1], sh->chroma_weight[1][ri[1]][1],
                        sh->chroma_offset[0][ri[0]][1], sh->chroma_offset[1][ri[1]][1]);
                } else {
                    weight_samples(oy, y[0], y[1], 16, m, 5, w0, w1, 0, 0);
                    weight_samples(ou, cu[0], cu[1], 4, m, 5, w0, w1, 0, 0);
                    weight_samples(ov, cv[0], cv[1], 4, m, 5, w0, w1, 0, 0);
                }
            } else {
                int l = ref[0] ? 0 : 1, m = mode == 1 ? 1 : 0;
                weight_samples(oy, y[l], NULL, 16, m, sh->luma_log2_wd, sh->luma_weight[l][ri[l]], 0, sh->luma_offset[l][ri[l]], 0);
                weight_samples(ou, cu[l], NULL, 4, m, sh->chroma_log2_wd, sh->chroma_weight[l][ri[l]][0], 0, sh->chroma_offset[l][ri[l]][0], 0);
                weight_samples(ov, cv[l], NULL, 4, m, sh->chroma_log2_wd, sh->chroma_weight[l][ri[l]][1], 0, sh->chroma_offset[l][ri[l]][1], 0);
            }
            int x0 = s->mb_x * 16 + bx * 4, y0 = s->mb_y * 16 + by * 4;
            for (int i = 0; i < 16; i++) pic->y[(y0 + (i >> 2)) * pic->stride_y + x0 + (i & 3)] = (uint8_t)oy[i];
            for (int i = 0; i < 4; i++) {
                pic->u[(y0 / 2 + (i >> 1)) * pic->stride_c + x0 / 2 + (i & 1)] = (uint8_t)ou[i];
                pic->v[(y0 / 2 + (i >> 1)) * pic->stride_c + x0 / 2 + (i & 1)] = (uint8_t)ov[i];
            }
        }
    }
    return 0;
}

/* ----------------------------- direct prediction (8.4.1.2) ---------------- */
static int min_positive(int a, int b) { return (a >= 0 && b >= 0) ? orc_min(a, b) : orc_max(a, b); }
/* Colocated 4x4 block (8.4.1.2.1): the motion that goes with block r of the current macroblock in RefPicList1[0].  Tables 8-6 and 8-8 without MBAFF:
 *   picture and colocated picture coded alike (frame / frame, field / field picture): the same macroblock, the same block (vertMvScale One_To_One);
 *   a FIELD picture whose RefPicList1[0] is a field of a FRAME picture (Frm_To_Fld): mbAddrCol6 = 2 * W * (CurrMbAddr / W) + CurrMbAddr % W + W * (yCol / 8)
 *     in the frame, yM = (2 * yCol) % 16;
 *   a FRAME picture whose RefPicList1[0] is a complementary field pair (Fld_To_Frm): the field nearer in order count (the top one only when strictly
 *     nearer), mbAddrCol7 = W * (CurrMbAddr / (2 * W)) + CurrMbAddr % W, yM = 8 * ((CurrMbAddr / W) % 2) + 4 * (yCol / 8).
 * (xCol, yCol) = position of block r; the mixed cases only occur with direct_8x8_inference_flag = 1, so r is a corner block.
 * *vscale: 0 / 1 (Frm_To_Fld) / 2 (Fld_To_Frm); *ref_pic_col: the identity of the referenced picture IN THE CURRENT PICTURE'S TERMS -- the field of the
 * current parity of the referenced frame (1), the frame of the referenced field (2) (8.4.1.2.3). */
static void colocated(Sl *s, int r, int *ref_col, int mv_col[2], int *ref_pic_col, int *vscale) {
    OrcDec *d = s->d;
    const Picture *l1 = d->ref_list[1][0], *st = l1->store ? l1->store : l1;
    const int W = d->mb_w, xCol = (r & 3) * 4, yCol = (r >> 2) * 4;
    const MbInfo *m; int rb = r;
    *vscale = 0;
    if (d->field_pic && !st->coded_fields) {
        m = &st->mbs[(2 * s->mb_y + yCol / 8) * W + s->mb_x];
        rb = (((2 * yCol) % 16) >> 2) * 4 + (xCol >> 2); *vscale = 1;
    } else if (!d->field_pic && st->coded_fields) {
        const int q = orc_abs(st->fpoc[0] - d->cur->poc) < orc_abs(st->fpoc[1] - d->cur->poc) ? 0 : 1;
        const int yM = 8 * (s->mb_y % 2) + 4 * (yCol / 8);
        m = &st->mbs[q * (W * (d->asps->mb_height / 2)) + (s->mb_y / 2) * W + s->mb_x];
        rb = (yM >> 2) * 4 + (xCol >> 2); *vscale = 2;
    } else m = &l1->mbs[s->mb_addr];
    int b8 = (rb >> 3) * 2 + ((rb & 3) >> 1);
    *ref_col = -1; mv_col[0] = mv_col[1] = 0; *ref_pic_col = -1;
    if (m->is_intra || m->slice_num < 0) return;
    int l = m->ref_idx[0][b8] >= 0 ? 0 : 1;
    if (m->ref_idx[l][b8] < 0) return;
    *ref_col = m->ref_idx[l][b8]; mv_col[0] = m->mv[l][rb][0]; mv_col[1] = m->mv[l][rb][1]; *ref_pic_col = m->ref_pic_id[l][b8];
    if (*vscale == 1) *ref_pic_col = 0x40000000 + 2 * *ref_pic_col + d->cur_parity;
    if (*vscale == 2) *ref_pic_col = (*ref_pic_col - 0x40000000) >> 1;
}
/* derive refIdx / mv of the 8x8 quadrants in `mask` (bit b8) by direct prediction; marks them in direct8 */
static int direct_pred(Sl *s, int mask) {
    MbInfo *mb = s->mb; OrcDec *d = s->d;
    if (d->ref_count[1] < 1 || !d->ref_list[1][0]) { snprintf(d->err, sizeof d->err, "direct prediction without RefPicList1[0]"); return -1; }
    int inf8 = s->sps->direct_8x8_inference;
    if (s->sh->direct_spatial_mv_pred) {
        int ref[2], mvp[2][2] = {{0, 0}, {0, 0}};
        for (int l = 0; l < 2; l++) {                          /* neighbours of the macroblock as one 16x16 partition */
            Nb A = nb_get(s, l, -1, 0), B = nb_get(s, l, 0, -1), C = nb_get(s, l, 4, -1);
            if (!C.avail) C = nb_get(s, l, -1, -1);
            ref[l] = min_positive(A.ref, min_positive(B.ref, C.ref));
        }
        int zero = ref[0] < 0 && ref[1] < 0;
        if (zero) ref[0] = ref[1] = 0;
        else for (int l = 0; l < 2; l++) if (ref[l] >= 0) {
            int saved = s->decoded_mask; s->decoded_mask = 0;  /* only neighbouring macroblocks take part (mbPartIdx 0 of a 16x16 partition) */
            predict_mv(s, l, 0, 0, 4, 4, ref[l], 0, 0, mvp[l]);
            s->decoded_mask = saved;
        }
        int col_short = d->ref_list[1][0]->is_ref != 2;
        for (int b8 = 0; b8 < 4; b8++) {
            if (!(mask & (1 << b8))) continue;
            mb->ref_idx[0][b8] = (int8_t)ref[0]; mb->ref_idx[1][b8] = (int8_t)ref[1];
            for (int k = 0; k < 4; k++) {
                int bx = (b8 & 1) * 2 + (k & 1), by = (b8 >> 1) * 2 + (k >> 1), r = by * 4 + bx;
                int rc = inf8 ? ((b8 >> 1) * 3) * 4 + (b8 & 1) * 3 : r;       /* corner block of the quadrant when direct_8x8_inference */
                int ref_col, mv_col[2], pid, vs;
                colocated(s, rc, &ref_col, mv_col, &pid, &vs);          /* (colZeroFlag: the vectors as stored, whatever vertMvScale is) */
                d->stats[ORC_ST_DIRECT_MIXED] += vs != 0;
                int col_zero = col_short && ref_col == 0 && mv_col[0] >= -1 && mv_col[0] <= 1 && mv_col[1] >= -1 && mv_col[1] <= 1;
                for (int l = 0; l < 2; l++) {
                    int z = zero || ref[l] < 0 || (ref[l] == 0 && col_zero);
                    mb->mv[l][r][0] = (int16_t)(z ? 0 : mvp[l][0]); mb->mv[l][r][1] = (int16_t)(z ? 0 : mvp[l][1]);
                }
            }
        }
    } else {
        for (int b8 = 0; b8 < 4; b8++) {
            if (!(mask & (1 << b8))) continue;
            for (int k = 0; k < 4; k++) {
                int bx = (b8 & 1) * 2 + (k & 1), by = (b8 >> 1) * 2 + (k >> 1), r = by * 4 + bx;
                int rc = inf8 ? ((b8 >> 1) * 3) * 4 + (b8 & 1) * 3 : r;
                int ref_col, mv_col[2], pid, ref0 = 0, vs;
                colocated(s, rc, &ref_col, mv_col, &pid, &vs);
                if (vs == 1) mv_col[1] = mv_col[1] / 2; else if (vs == 2) mv_col[1] *= 2;      /* 8.4.1.2.3 ("/": towards zero) */
                d->stats[ORC_ST_DIRECT_MIXED] += vs != 0;
                if (ref_col >= 0) {
                    ref0 = -1;
                    for (int i = 0; i < d->ref_count[0]; i++) if (d->ref_list[0][i] && d->ref_list[0][i]->id == pid) { ref0 = i; break; }
                    if (ref0 < 0) { snprintf(d->err, sizeof d->err, "temporal direct: colocated reference is not in RefPicList0"); return -1; }
                }
                const Picture *p0 = d->ref_list[0][ref0], *p1 = d->ref_list[1][0];
                if (!p0) { snprintf(d->err, sizeof d->err, "temporal direct without RefPicList0[%d]", ref0); return -1; }
                int mv0[2], mv1[2];
                int tb = orc_clip3(-128, 127, d->cur->poc - p0->poc), td = orc_clip3(-128, 127, p1->poc - p0->poc);
                if (p0->is_ref == 2 || td == 0) { mv0[0] = mv_col[0]; mv0[1] = mv_col[1]; mv1[0] = mv1[1] = 0; }
                else {
                    int tx = (16384 + orc_abs(td / 2)) / td, dsf = orc_clip3(-1024, 1023, (tb * tx + 32) >> 6);
                    for (int c = 0; c < 2; c++) { mv0[c] = (dsf * mv_col[c] + 128) >> 8; mv1[c] = mv0[c] - mv_col[c]; }
                }
                mb->ref_idx[0][b8] = (int8_t)ref0; mb->ref_idx[1][b8] = 0;
                mb->mv[0][r][0] = (int16_t)mv0[0]; mb->mv[0][r][1] = (int16_t)mv0[1];
                mb->mv[1][r][0] = (int16_t)mv1[0]; mb->mv[1][r][1] = (int16_t)mv1[1];
            }
        }
    }
    mb->direct8 |= (uint8_t)mask;
    return 0;
}
static void mark_decoded8(Sl *s, int b8) {
    int bx = (b8 & 1) * 2, by = (b8 >> 1) * 2;
    s->decoded_mask |= (1 << (by * 4 + bx)) | (1 << (by * 4 + bx + 1)) | (1 << (by * 4 + bx + 4)) | (1 << (by * 4 + bx + 5));
}

/* ----------------------------- residual add ------------------------------ */
static void recon_chroma_residual(Sl *s, int cbp) {
    Picture *pic = s->pic; MbInfo *mb = s->mb;
    if (!(cbp & 0x30)) return;
    for (int pl = 0; pl < 2; pl++) {
        uint8_t *dst = (pl ? pic->v : pic->u) + (s->mb_y * 8) * pic->stride_c + s->mb_x * 8;
        int qpc = mb->qpc[pl];
        const uint8_t *list = s->pps->scaling4[(mb->is_intra ? 0 : 3) + 1 + pl];
        /* 8.5.11.1/2: 2x2 chroma DC transform + scaling */
        const int16_t *c = s->cdc[pl];
        int f[4] = { c[0] + c[1] + c[2] + c[3], c[0] - c[1] + c[2] - c[3], c[0] + c[1] - c[2] - c[3], c[0] - c[1] - c[2] + c[3] };
        int ls0 = level_scale4(list, qpc % 6, 0);
        for (int k = 0; k < 4; k++) {
            int dc = ((f[k] * ls0) << (qpc / 6)) >> 5;
            int16_t blk[16]; int dq[16], r[16];
            memcpy(blk, s->cac[pl][k], sizeof blk);
            blk[0] = 0;
            scale4x4(blk, dq, list, qpc, 0);
            dq[0] = dc;
            idct4x4(dq, r);
            add_block(dst + (k >> 1) * 4 * pic->stride_c + (k & 1) * 4, pic->stride_c, r, 4);
        }
    }
}


/* ----------------------------- Intra 8x8 (8.3.2) ------------------------- */
/* reference sample filtering 8.3.2.2.1 + the nine modes 8.3.2.2.2-10 for 8x8 block b8 */
static int pred8x8l(Sl *s, int b8, int mode, uint8_t *dst, int stride) {
    int bx = b8 & 1, by = b8 >> 1;
    MbInfo *mA = bx ? s->mb : mb_at(s, s->mb_x - 1, s->mb_y);
    MbInfo *mB = by ? s->mb : mb_at(s, s->mb_x, s->mb_y - 1);
    MbInfo *mD = (bx && by) ? s->mb : (bx ? mb_at(s, s->mb_x, s->mb_y - 1) : (by ? mb_at(s, s->mb_x - 1, s->mb_y) : mb_at(s, s->mb_x - 1, s->mb_y - 1)));
    int availA = bx || intra_usable(s, mA), availB = by || intra_usable(s, mB), availD = (bx && by) || intra_usable(s, mD), availC;
    if (b8 == 0) availC = intra_usable(s, mb_at(s, s->mb_x, s->mb_y - 1));
    else if (b8 == 1) availC = intra_usable(s, mb_at(s, s->mb_x + 1, s->mb_y - 1));
    else availC = b8 == 2;
    int t[16], l[8], c = 128, ft[16], fl[8], fc = 128;       /* raw p[x,-1], p[-1,y], p[-1,-1] and the filtered p' */
    for (int i = 0; i < 8; i++) { t[i] = availB ? dst[-stride + i] : 128; l[i] = availA ? dst[i * stride - 1] : 128; }
    for (int i = 8; i < 16; i++) t[i] = availB ? (availC ? dst[-stride + i] : t[7]) : 128;
    if (availD) c = dst[-stride - 1];
    if (availB) {
        ft[0] = availD ? (c + 2 * t[0] + t[1] + 2) >> 2 : (3 * t[0] + t[1] + 2) >> 2;
        for (int i = 1; i < 15; i++) ft[i] = (t[i - 1] + 2 * t[i] + t[i + 1] + 2) >> 2;
        ft[15] = (t[14] + 3 * t[15] + 2) >> 2;
    } else for (int i = 0; i < 16; i++) ft[i] = 128;
    if (availD) {
        if (availA && availB) fc = (t[0] + 2 * c + l[0] + 2) >> 2;
        else if (availB) fc = (3 * c + t[0] + 2) >> 2;
        else if (availA) fc = (3 * c + l[0] + 2) >> 2;
        else fc = c;
    }
    if (availA) {
        fl[0] = availD ? (c + 2 * l[0] + l[1] + 2) >> 2 : (3 * l[0] + l[1] + 2) >> 2;
        for (int i = 1; i < 7; i++) fl[i] = (l[i - 1] + 2 * l[i] + l[i + 1] + 2) >> 2;
        fl[7] = (l[6] + 3 * l[7] + 2) >> 2;
    } else for (int i = 0; i < 8; i++) fl[i] = 128;
    if (((mode == 0 || mode == 3 || mode == 7) && !availB) || ((mode == 1 || mode == 8) && !availA) ||
        ((mode == 4 || mode == 5 || mode == 6) && !(availA && availB && availD))) return -1;
#define T(i) ((i) < 0 ? fc : ft[i])
#define L(i) ((i) < 0 ? fc : fl[i])
    int p[64];
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
        int v;
        switch (mode) {
        case 0: v = ft[x]; break;
        case 1: v = fl[y]; break;
        case 2: {
            int st = 0, sl = 0;
            for (int i = 0; i < 8; i++) { st += ft[i]; sl += fl[i]; }
            v = (availA && availB) ? (st + sl + 8) >> 4 : availA ? (sl + 4) >> 3 : availB ? (st + 4) >> 3 : 128;
            break; }
        case 3: v = (x == 7 && y == 7) ? (ft[14] + 3 * ft[15] + 2) >> 2 : (ft[x + y] + 2 * ft[x + y + 1] + ft[x + y + 2] + 2) >> 2; break;
        case 4:
            if (x > y) v = (T(x - y - 2) + 2 * T(x - y - 1) + T(x - y) + 2) >> 2;
            else if (x < y) v = (L(y - x - 2) + 2 * L(y - x - 1) + L(y - x) + 2) >> 2;
            else v = (ft[0] + 2 * fc + fl[0] + 2) >> 2;
            break;
        case 5: {
            int z = 2 * x - y, i = x - (y >> 1);
            if (z >= 0 && !(z & 1)) v = (T(i - 1) + T(i) + 1) >> 1;
            else if (z >= 0) v = (T(i - 2) + 2 * T(i - 1) + T(i) + 2) >> 2;
            else if (z == -1) v = (fl[0] + 2 * fc + ft[0] + 2) >> 2;
            else v = (L(y - 2 * x - 1) + 2 * L(y - 2 * x - 2) + L(y - 2 * x - 3) + 2) >> 2;
            break; }
        case 6: {
            int z = 2 * y - x, i = y - (x >> 1);
            if (z >= 0 && !(z & 1)) v = (L(i - 1) + L(i) + 1) >> 1;
            else if (z >= 0) v = (L(i - 2) + 2 * L(i - 1) + L(i) + 2) >> 2;
            else if (z == -1) v = (fl[0] + 2 * fc + ft[0] + 2) >> 2;
            else v = (T(x - 2 * y - 1) + 2 * T(x - 2 * y - 2) + T(x - 2 * y - 3) + 2) >> 2;
            break; }
        case 7: {
            int i = x + (y >> 1);
            v = (y & 1) ? (ft[i] + 2 * ft[i + 1] + ft[i + 2] + 2) >> 2 : (ft[i] + ft[i + 1] + 1) >> 1;
            break; }
        default: {
            int z = x + 2 * y, i = y + (x >> 1);
            if (z > 13) v = fl[7];
            else if (z == 13) v = (fl[6] + 3 * fl[7] + 2) >> 2;
            else if (z & 1) v = (fl[i] + 2 * fl[i + 1] + fl[i + 2] + 2) >> 2;
            else v = (fl[i] + fl[i + 1] + 1) >> 1;
            break; }
        }
        p[y * 8 + x] = v;
    }
#undef T
#undef L
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) dst[y * stride + x] = (uint8_t)p[y * 8 + x];
    return 0;
}
/* 8.5.13: scaling + inverse transform of one 8x8 residual block, added to dst */
static void recon8x8(Sl *s, int b8, const uint8_t *list, int qp, uint8_t *dst, int stride) {
    int dq[64], res[64];
    for (int k = 0; k < 64; k++) {
        int ls = level_scale8(list, qp % 6, k), c = s->luma8[b8][k];
        dq[k] = qp >= 36 ? (c * ls) << (qp / 6 - 6) : (c * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
    }
    idct8x8(dq, res);
    add_block(dst, stride, res, 8);
}

/* ----------------------------- syntax digest ------------------------------ */
/* FNV-1a over a canonical serialisation of the parsed macroblock, used by
 * tests/test_host_parser.py to compare the product's host entropy decoder with
 * this oracle without needing a GPU.  Layout mirrors struct Canon in
 * jmcodec_amd/csrc/h264_cavlc.cpp (little-endian host). */
static void digest_mb(Sl *s) {
    OrcDec *d = s->d; MbInfo *mb = s->mb;
    if (!d->digest_on) return;
    uint8_t buf[976]; size_t n = 0;
    memset(buf, 0, sizeof buf);
    uint32_t addr = (uint32_t)s->mb_addr; memcpy(buf, &addr, 4); n = 4;
    int kind = !mb->is_intra ? 0 : (mb->is_pcm ? 3 : (mb->is_i16 ? 2 : 1));
    buf[n++] = (uint8_t)(kind | (mb->t8x8 ? 16 : 0)); buf[n++] = mb->qp;
    buf[n++] = (uint8_t)((kind == 1 || kind == 2) ? s->chroma_pred_mode : 0);
    buf[n++] = (uint8_t)(kind == 2 ? s->i16_pred_mode : 0);
    if (kind == 1) memcpy(buf + n, mb->i4mode, 16);
    n += 16;
    for (int i = 0; i < 4; i++) buf[n++] = (uint8_t)(kind == 0 ? mb->ref_idx[0][i] : -1);
    if (kind == 0) memcpy(buf + n, mb->mv[0], 64);
    n += 64;
    if (kind != 3) {
        memcpy(buf + n, s->i16dc, 32);
        if (mb->t8x8) memcpy(buf + n + 32, s->luma8, 512); else memcpy(buf + n + 32, s->luma, 512);
        memcpy(buf + n + 544, s->cdc, 16); memcpy(buf + n + 560, s->cac, 256);
    }
    n += 816;
    for (int i = 0; i < 4; i++) buf[n++] = (uint8_t)(kind == 0 ? mb->ref_idx[1][i] : -1);
    if (kind == 0) memcpy(buf + n, mb->mv[1], 64);
    n += 64;
    uint64_t h = d->digest;
    for (size_t i = 0; i < n; i++) { h ^= buf[i]; h *= 1099511628211ull; }
    d->digest = h; d->digest_mbs++;
}

/* ----------------------------- macroblock layer -------------------------- */
static void mb_reset(Sl *s) {
    MbInfo *mb = s->mb;
    memset(mb, 0, sizeof *mb);
    mb->slice_num = (int16_t)s->d->slice_num;
    for (int l = 0; l < 2; l++) for (int i = 0; i < 4; i++) { mb->ref_idx[l][i] = -1; mb->ref_pic_id[l][i] = -1; }
    memset(mb->i4mode, 2, 16);
    mb->disable_deblock = (uint8_t)s->sh->disable_deblock;
    mb->alpha_off = (int8_t)s->sh->alpha_c0_offset; mb->beta_off = (int8_t)s->sh->beta_offset;
    memset(s->luma, 0, sizeof s->luma); memset(s->luma8, 0, sizeof s->luma8);
    memset(s->i16dc, 0, sizeof s->i16dc); memset(s->cdc, 0, sizeof s->cdc); memset(s->cac, 0, sizeof s->cac);
    s->decoded_mask = 0;
}
static void mb_set_qp(Sl *s) {
    s->mb->qp = (uint8_t)s->qp;
    s->mb->qpc[0] = (uint8_t)chroma_qp(s->pps, s->qp, 0);
    s->mb->qpc[1] = (uint8_t)chroma_qp(s->pps, s->qp, 1);
}

static int decode_skip_mb(Sl *s) {
    mb_reset(s);
    MbInfo *mb = s->mb;
    mb->is_skip = 1; mb->mb_type_p = 0;
    mb_set_qp(s);
    s->last_dqp_nonzero = 0;
    if (s->sh->slice_type == SLICE_B) {                       /* B_Skip: direct prediction, no residual */
        s->d->stats[ORC_ST_BSKIP]++;
        mb->b_direct16 = 1;
        if (direct_pred(s, 15) < 0) return -1;
        digest_mb(s);
        return inter_recon(s);
    }
    s->d->stats[ORC_ST_PSKIP]++;
    int mvp[2] = {0, 0};
    MbInfo *mA = mb_at(s, s->mb_x - 1, s->mb_y), *mB = mb_at(s, s->mb_x, s->mb_y - 1);
    if (mA && mB) {
        Nb A = nb_get(s, 0, -1, 0), B = nb_get(s, 0, 0, -1);
        if (!((A.ref == 0 && A.mv[0] == 0 && A.mv[1] == 0) || (B.ref == 0 && B.mv[0] == 0 && B.mv[1] == 0)))
            predict_mv(s, 0, 0, 0, 4, 4, 0, 0, 0, mvp);
    }
    for (int i = 0; i < 4; i++) mb->ref_idx[0][i] = 0;
    set_mv(s, 0, 0, 0, 4, 4, mvp[0], mvp[1]);
    digest_mb(s);
    return inter_recon(s);
}

/* entropy-mode dispatch of the 7.3.5 descriptors: ue(v)|ae(v), te(v)|ae(v), se(v)|ae(v), me(v)|ae(v) */
static int rd_ref_idx(Sl *s, int list, int bx, int by, int nref) {
    if (s->cabac_on) return orc_cabac_ref_idx(s, list, bx, by);
    return bits_te(s->b, nref - 1);
}
static void rd_mvd(Sl *s, int list, int bx, int by, int bw, int bh, int mvd[2]) {
    if (s->cabac_on) { mvd[0] = orc_cabac_mvd(s, list, bx, by, 0); mvd[1] = orc_cabac_mvd(s, list, bx, by, 1); }
    else { mvd[0] = bits_se(s->b); mvd[1] = bits_se(s->b); }
    for (int y = by; y < by + bh; y++) for (int x = bx; x < bx + bw; x++) {
        s->mb->mvd[list][y * 4 + x][0] = (uint16_t)orc_min(orc_abs(mvd[0]), 65535);
        s->mb->mvd[list][y * 4 + x][1] = (uint16_t)orc_min(orc_abs(mvd[1]), 65535);
    }
}

/* 7.3.5.3 residual() with CABAC */
static int parse_residual_cabac(Sl *s, int cbp) {
    MbInfo *mb = s->mb;
    int16_t tmp[64];
    const uint8_t *z4 = s->d->field_pic ? orc_fieldscan4 : orc_zigzag4, *z8 = s->d->field_pic ? orc_fieldscan8 : orc_zigzag8;
    if (mb->is_i16) {
        if (orc_cabac_residual_block(s, 0, 0, tmp, 16) < 0) return -1;
        for (int i = 0; i < 16; i++) s->i16dc[z4[i]] = tmp[i];
    }
    for (int b8 = 0; b8 < 4; b8++) {
        if (!(cbp & (1 << b8))) continue;
        if (mb->t8x8) {
            int n = orc_cabac_residual_block(s, 5, b8, tmp, 64);
            if (n < 0) return -1;
            for (int i = 0; i < 64; i++) s->luma8[b8][z8[i]] = tmp[i];
            for (int k = 0; k < 4; k++) {
                int r = ((b8 >> 1) * 2 + (k >> 1)) * 4 + (b8 & 1) * 2 + (k & 1);
                mb->total_coeff[r] = (uint8_t)orc_min(n, 16);
                mb->cbf |= 1u << r;                 /* 7.4.5.3.3: coded_block_flag of an 8x8 block is inferred to be 1 */
            }
            continue;
        }
        for (int k = 0; k < 4; k++) {
            int blk = b8 * 4 + k, bx = blk_x(blk), by = blk_y(blk), r = by * 4 + bx, n;
            if (mb->is_i16) {
                n = orc_cabac_residual_block(s, 1, r, tmp, 15);
                if (n < 0) return -1;
                for (int i = 0; i < 15; i++) s->luma[r][z4[i + 1]] = tmp[i];
            } else {
                n = orc_cabac_residual_block(s, 2, r, tmp, 16);
                if (n < 0) return -1;
                for (int i = 0; i < 16; i++) s->luma[r][z4[i]] = tmp[i];
            }
            mb->total_coeff[r] = (uint8_t)n;
        }
    }
    if (cbp & 0x30)
        for (int pl = 0; pl < 2; pl++)
            if (orc_cabac_residual_block(s, 3, pl, s->cdc[pl], 4) < 0) return -1;
    if (cbp & 0x20)
        for (int pl = 0; pl < 2; pl++)
            for (int k = 0; k < 4; k++) {
                int n = orc_cabac_residual_block(s, 4, pl * 4 + k, tmp, 15);
                if (n < 0) return -1;
                for (int i = 0; i < 15; i++) s->cac[pl][k][z4[i + 1]] = tmp[i];
                mb->total_coeff[16 + 4 * pl + k] = (uint8_t)n;
            }
    return 0;
}

/* 8.3.1.1 / 8.3.2.1: predicted Intra4x4 / Intra8x8 mode for the block whose top-left 4x4 is (bx,by) */
static int pred_intra_mode(Sl *s, int bx, int by) {
    MbInfo *mb = s->mb;
    MbInfo *mA = bx > 0 ? mb : mb_at(s, s->mb_x - 1, s->mb_y);
    MbInfo *mB = by > 0 ? mb : mb_at(s, s->mb_x, s->mb_y - 1);
    if (!mA || !mB) return 2;
    if ((!mA->is_intra || !mB->is_intra) && s->pps->constrained_intra_pred) return 2;
    /* neighbour not coded as I4x4 / I8x8 -> DC; modes of I8x8 macroblocks are stored replicated over their 4x4 blocks */
    int modeA = (mA->is_intra && !mA->is_i16 && !mA->is_pcm) ? (bx > 0 ? mb->i4mode[by * 4 + bx - 1] : mA->i4mode[by * 4 + 3]) : 2;
    int modeB = (mB->is_intra && !mB->is_i16 && !mB->is_pcm) ? (by > 0 ? mb->i4mode[(by - 1) * 4 + bx] : mB->i4mode[12 + bx]) : 2;
    return orc_min(modeA, modeB);
}

static int decode_mb(Sl *s) {
    Bits *b = s->b; Picture *pic = s->pic; const SliceHdr *sh = s->sh;
    const int cab = s->cabac_on;
    mb_reset(s);
    MbInfo *mb = s->mb;
    int mb_type = cab ? orc_cabac_mb_type(s) : (int)bits_ue(b);
    int is_intra_type = -1;                /* I-slice numbering when intra */
    if (sh->slice_type == SLICE_I) is_intra_type = mb_type;
    else if (sh->slice_type == SLICE_P) { if (mb_type >= 5) is_intra_type = mb_type - 5; }
    else if (mb_type >= 23) is_intra_type = mb_type - 23;
    if (is_intra_type > 25 || mb_type > 48) { snprintf(s->d->err, sizeof s->d->err, "bad mb_type %d", mb_type); return -1; }

    uint8_t *dy = pic->y + (s->mb_y * 16) * pic->stride_y + s->mb_x * 16;
    uint8_t *du = pic->u + (s->mb_y * 8) * pic->stride_c + s->mb_x * 8;
    uint8_t *dv = pic->v + (s->mb_y * 8) * pic->stride_c + s->mb_x * 8;

    if (is_intra_type == 25) {             /* I_PCM, 7.3.5 */
        mb->is_intra = 1; mb->is_pcm = 1;
        s->d->stats[ORC_ST_PCM]++;
        if (cab) b->pos = (b->pos + 7) & ~(size_t)7;
        /* the 9 bits read ahead end exactly with the last bit of the encoder's flush (9.3.4.5): only the pcm_alignment_zero_bits remain */
        else while (!bits_aligned(b)) if (bits_u1(b)) { snprintf(s->d->err, sizeof s->d->err, "pcm_alignment_zero_bit != 0"); return -1; }
        for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) dy[y * pic->stride_y + x] = (uint8_t)bits_u(b, 8);
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) du[y * pic->stride_c + x] = (uint8_t)bits_u(b, 8);
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) dv[y * pic->stride_c + x] = (uint8_t)bits_u(b, 8);
        memset(mb->total_coeff, 16, sizeof mb->total_coeff);
        mb->cbf = 0x7FFFFFF;
        mb->qp = 0;                        /* 8.7.2.2: qPp = 0 for I_PCM in the deblocking filter */
        mb->qpc[0] = (uint8_t)chroma_qp(s->pps, 0, 0); mb->qpc[1] = (uint8_t)chroma_qp(s->pps, 0, 1);
        mb->cbp = 0x2f;
        s->last_dqp_nonzero = 0;
        if (cab && orc_cabac_init_engine(&s->c, b) < 0) return -1;
        digest_mb(s);
        return b->err ? -1 : 0;
    }

    int cbp = 0;
    if (is_intra_type >= 0) {
        mb->is_intra = 1;
        if (is_intra_type == 0) {          /* I_NxN */
            if (s->pps->transform_8x8_mode) mb->t8x8 = (uint8_t)(cab ? orc_cabac_transform8x8_flag(s) : (int)bits_u1(b));
            int nblk = mb->t8x8 ? 4 : 16;
            for (int blk = 0; blk < nblk; blk++) {
                int bx = mb->t8x8 ? (blk & 1) * 2 : blk_x(blk), by = mb->t8x8 ? (blk >> 1) * 2 : blk_y(blk);
                int pred = pred_intra_mode(s, bx, by), mode;
                if (cab) { int rem = orc_cabac_intra_pred_mode(s); mode = rem < 0 ? pred : (rem < pred ? rem : rem + 1); }
                else if (bits_u1(b)) mode = pred;
                else { int rem = bits_u(b, 3); mode = rem < pred ? rem : rem + 1; }
                if (mb->t8x8) { mb->i4mode[by * 4 + bx] = mb->i4mode[by * 4 + bx + 1] = mb->i4mode[by * 4 + bx + 4] = mb->i4mode[by * 4 + bx + 5] =
                    (uint8_t)mode; }
                else mb->i4mode[by * 4 + bx] = (uint8_t)mode;
            }
        } else {                           /* I_16x16: Table 7-11 */
            int k = is_intra_type - 1;
            mb->is_i16 = 1;
            s->i16_pred_mode = k % 4;
            cbp = ((k / 4) % 3) << 4 | (k >= 12 ? 15 : 0);
        }
        s->chroma_pred_mode = cab ? orc_cabac_chroma_pred_mode(s) : (int)bits_ue(b);
        if (s->chroma_pred_mode > 3) { snprintf(s->d->err, sizeof s->d->err, "bad intra_chroma_pred_mode"); return -1; }
        mb->chroma_pred_mode = (uint8_t)s->chroma_pred_mode;
    } else {
        /* ---- P / B macroblock: 7.3.5.1 mb_pred / 7.3.5.2 sub_mb_pred ---- */
        static const uint8_t b_pair[9][2] = {{0, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 2}, {1, 2}, {2, 0}, {2, 1}, {2, 2}};   /* Table 7-14: 0 L0 1 L1 2 Bi */
        static const uint8_t b_sub_pred[13] = {3, 0, 1, 2, 0, 0, 1, 1, 2, 2, 0, 1, 2};                                   /* Table 7-18: 3 direct */
        static const uint8_t b_sub_shape[13] = {0, 0, 0, 0, 1, 2, 1, 2, 1, 2, 3, 3, 3};                                  /* 0 8x8 1 8x4 2 4x8 3 4x4 */
        const int is_b = sh->slice_type == SLICE_B;
        int shape, pred[4] = {0, 0, 0, 0};   /* shape 0 16x16 1 16x8 2 8x16 3 8x8; pred per partition / sub-macroblock */
        int sub_shape[4] = {0, 0, 0, 0};
        if (!is_b) { shape = mb_type > 3 ? 3 : mb_type; }
        else if (mb_type == 0) shape = 4;                                   /* B_Direct_16x16 */
        else if (mb_type <= 3) { shape = 0; pred[0] = mb_type - 1; }
        else if (mb_type == 22) shape = 3;
        else { shape = (mb_type & 1) ? 2 : 1; pred[0] = b_pair[(mb_type - 4) >> 1][0]; pred[1] = b_pair[(mb_type - 4) >> 1][1]; }
        mb->mb_type_p = (uint8_t)(shape & 3);
        if (shape == 4) {
            mb->b_direct16 = 1; s->d->stats[ORC_ST_BDIRECT]++;
            if (direct_pred(s, 15) < 0) return -1;
            if (!s->sps->direct_8x8_inference) mb->mb_type_p |= 4;          /* no 8x8 transform without direct_8x8_inference */
        } else if (shape <= 2) {
            int nparts = shape == 0 ? 1 : 2, refs[2][2] = {{-1, -1}, {-1, -1}};
            if (is_b) s->d->stats[ORC_ST_BINTER]++;
            for (int l = 0; l < (is_b ? 2 : 1); l++) {
                int nref = sh->num_ref_idx[l];
                for (int p = 0; p < nparts; p++) {
                    if (!(pred[p] == 2 || pred[p] == l)) continue;
                    int bx = shape == 2 ? p * 2 : 0, by = shape == 1 ? p * 2 : 0;
                    int bw = shape == 2 ? 2 : 4, bh = shape == 1 ? 2 : 4;
                    refs[l][p] = 0;
                    if (nref > 1) { refs[l][p] = rd_ref_idx(s, l, bx, by, nref); if (refs[l][p] < 0 || refs[l][p] >= nref) return -1; }
                    for (int y = by; y < by + bh; y += 2) for (int x = bx; x < bx + bw; x += 2) mb->ref_idx[l][(y >> 1) * 2 + (x >> 1)] = (int8_t)refs[l][p];
                }
            }
            for (int l = 0; l < (is_b ? 2 : 1); l++) {
                s->decoded_mask = 0;
                for (int p = 0; p < nparts; p++) {
                    int bx = shape == 2 ? p * 2 : 0, by = shape == 1 ? p * 2 : 0;
                    int bw = shape == 2 ? 2 : 4, bh = shape == 1 ? 2 : 4;
                    if (refs[l][p] < 0) { for (int y = by; y < by + bh; y++) for (int x = bx; x < bx + bw; x++) s->decoded_mask |= 1 << (y * 4 + x); continue; }
                    int mvp[2], mvd[2]; predict_mv(s, l, bx, by, bw, bh, refs[l][p], shape, p, mvp);
                    rd_mvd(s, l, bx, by, bw, bh, mvd);
                    set_mv(s, l, bx, by, bw, bh, mvp[0] + mvd[0], mvp[1] + mvd[1]);
                }
            }
        } else {
            int sub[4], refs[2][4] = {{-1, -1, -1, -1}, {-1, -1, -1, -1}}, dmask = 0;
            if (is_b) s->d->stats[ORC_ST_BINTER]++;
            for (int i = 0; i < 4; i++) {
                sub[i] = cab ? orc_cabac_sub_mb_type(s) : (int)bits_ue(b);
                if (sub[i] > (is_b ? 12 : 3)) return -1;
                if (is_b) { pred[i] = b_sub_pred[sub[i]]; sub_shape[i] = b_sub_shape[sub[i]]; if (pred[i] == 3) dmask |= 1 << i; }
                else { pred[i] = 0; sub_shape[i] = sub[i]; }
                if (sub_shape[i] != 0 || (pred[i] == 3 && !s->sps->direct_8x8_inference)) mb->mb_type_p |= 4;   /* noSubMbPartSizeLessThan8x8Flag = 0 */
            }
            /* direct sub-macroblocks: derived from neighbouring macroblocks / the colocated picture only */
            if (dmask && direct_pred(s, dmask) < 0) return -1;
            for (int l = 0; l < (is_b ? 2 : 1); l++) {
                int nref = sh->num_ref_idx[l];
                for (int i = 0; i < 4; i++) {
                    if (!(pred[i] == 2 || pred[i] == l)) continue;
                    refs[l][i] = 0;
                    if (nref > 1 && !(!is_b && mb_type == 4)) { refs[l][i] = rd_ref_idx(s, l, (i & 1) * 2, (i >> 1) * 2, nref);
                        if (refs[l][i] < 0 || refs[l][i] >= nref) return -1; }
                    mb->ref_idx[l][i] = (int8_t)refs[l][i];
                }
            }
            for (int l = 0; l < (is_b ? 2 : 1); l++) {
                s->decoded_mask = 0;
                for (int i = 0; i < 4; i++) {
                    int ox = (i & 1) * 2, oy = (i >> 1) * 2, sp = sub_shape[i];
                    if (refs[l][i] < 0) { mark_decoded8(s, i); continue; }   /* direct, or this list unused: values are in place */
                    int nsp = sp == 0 ? 1 : (sp == 3 ? 4 : 2);
                    int bw = (sp == 0 || sp == 1) ? 2 : 1, bh = (sp == 0 || sp == 2) ? 2 : 1;
                    for (int p = 0; p < nsp; p++) {
                        int bx = ox + (sp == 1 ? 0 : (sp == 2 ? p : (p & 1)));
                        int by = oy + (sp == 1 ? p : (sp == 2 ? 0 : (p >> 1)));
                        int mvp[2], mvd[2]; predict_mv(s, l, bx, by, bw, bh, refs[l][i], 0, 0, mvp);
                        rd_mvd(s, l, bx, by, bw, bh, mvd);
                        set_mv(s, l, bx, by, bw, bh, mvp[0] + mvd[0], mvp[1] + mvd[1]);
                    }
                }
            }
        }
    }
    if (!mb->is_i16) {
        if (cab) cbp = orc_cabac_cbp(s);
        else {
            unsigned code = bits_ue(b);
            if (code > 47) { snprintf(s->d->err, sizeof s->d->err, "bad coded_block_pattern"); return -1; }
            cbp = mb->is_intra ? orc_cbp_intra[code] : orc_cbp_inter[code];
        }
        /* 7.3.5: transform_size_8x8_flag of a non-intra macroblock (noSubMbPartSizeLessThan8x8Flag) */
        if ((cbp & 15) && s->pps->transform_8x8_mode && !mb->is_intra && !(mb->mb_type_p & 4))
            mb->t8x8 = (uint8_t)(cab ? orc_cabac_transform8x8_flag(s) : (int)bits_u1(b));
    }
    {
        long *st = s->d->stats;
        if (mb->is_intra) st[mb->is_i16 ? ORC_ST_I16 : (mb->t8x8 ? ORC_ST_I8 : ORC_ST_I4)]++;
        else {
            if (sh->slice_type == SLICE_P) st[ORC_ST_P16 + (mb->mb_type_p & 3)]++;
            if (mb->mb_type_p & 4) st[ORC_ST_SUB_SMALL]++;
            if (mb->t8x8) st[ORC_ST_T8_INTER]++;
            if (mb->ref_idx[0][0] > 0 || mb->ref_idx[0][1] > 0 || mb->ref_idx[0][2] > 0 || mb->ref_idx[0][3] > 0) st[ORC_ST_MULTIREF]++;
        }
    }
    mb->mb_type_p &= 3;
    mb->cbp = (uint16_t)cbp;
    if (cbp > 0 || mb->is_i16) {
        int dqp = cab ? orc_cabac_qp_delta(s) : bits_se(b);
        if (dqp < -26 || dqp > 25) { snprintf(s->d->err, sizeof s->d->err, "mb_qp_delta out of range"); return -1; }
        s->qp = (s->qp + dqp + 52) % 52;
        s->last_dqp_nonzero = dqp != 0;
    } else s->last_dqp_nonzero = 0;
    mb_set_qp(s);
    if (cbp > 0 || mb->is_i16) {
        if ((cab ? parse_residual_cabac(s, cbp) : parse_residual(s, cbp)) < 0) {
            if (!s->d->err[0]) snprintf(s->d->err, sizeof s->d->err, "%s residual error at MB %d", cab ? "CABAC" : "CAVLC", s->mb_addr);
            return -1;
        }
    }
    if (b->err) return -1;
    digest_mb(s);

    /* ------------------------- reconstruction ------------------------- */
    int qp = s->qp;
    if (!mb->is_intra) {
        if (inter_recon(s) < 0) return -1;
        if (mb->t8x8) {
            for (int b8 = 0; b8 < 4; b8++)
                if (cbp & (1 << b8)) recon8x8(s, b8, s->pps->scaling8[1], qp, dy + (b8 >> 1) * 8 * pic->stride_y + (b8 & 1) * 8, pic->stride_y);
        } else {
            const uint8_t *list = s->pps->scaling4[3];
            for (int r = 0; r < 16; r++) {
                if (!mb->total_coeff[r]) continue;
                int dq[16], res[16];
                scale4x4(s->luma[r], dq, list, qp, 0);
                idct4x4(dq, res);
                add_block(dy + (r >> 2) * 4 * pic->stride_y + (r & 3) * 4, pic->stride_y, res, 4);
            }
        }
    } else if (mb->is_i16) {
        if (pred16x16(s, s->i16_pred_mode, dy, pic->stride_y) < 0) { snprintf(s->d->err, sizeof s->d->err, "Intra16x16 mode needs unavailable neighbour");
            return -1; }
        const uint8_t *list = s->pps->scaling4[0];
        /* 8.5.10: 4x4 luma DC Hadamard + scaling */
        int f[16], g[16];
        const int16_t *c = s->i16dc;
        for (int i = 0; i < 4; i++) {
            int a0 = c[4 * i] + c[4 * i + 1] + c[4 * i + 2] + c[4 * i + 3], a1 = c[4 * i] + c[4 * i + 1] - c[4 * i + 2] - c[4 * i + 3];
            int a2 = c[4 * i] - c[4 * i + 1] - c[4 * i + 2] + c[4 * i + 3], a3 = c[4 * i] - c[4 * i + 1] + c[4 * i + 2] - c[4 * i + 3];
            f[4 * i] = a0; f[4 * i + 1] = a1; f[4 * i + 2] = a2; f[4 * i + 3] = a3;
        }
        for (int j = 0; j < 4; j++) {
            g[j] = f[j] + f[4 + j] + f[8 + j] + f[12 + j]; g[4 + j] = f[j] + f[4 + j] - f[8 + j] - f[12 + j];
            g[8 + j] = f[j] - f[4 + j] - f[8 + j] + f[12 + j]; g[12 + j] = f[j] - f[4 + j] + f[8 + j] - f[12 + j];
        }
        int ls0 = level_scale4(list, qp % 6, 0);
        for (int r = 0; r < 16; r++) {
            int dc = qp >= 36 ? (g[r] * ls0) << (qp / 6 - 6) : (g[r] * ls0 + (1 << (5 - qp / 6))) >> (6 - qp / 6);
            int dq[16], res[16];
            scale4x4(s->luma[r], dq, list, qp, 0);
            dq[0] = dc;
            idct4x4(dq, res);
            add_block(dy + (r >> 2) * 4 * pic->stride_y + (r & 3) * 4, pic->stride_y, res, 4);
        }
    } else if (mb->t8x8) {                 /* Intra 8x8: predict + residual per 8x8 block */
        for (int b8 = 0; b8 < 4; b8++) {
            uint8_t *dst = dy + (b8 >> 1) * 8 * pic->stride_y + (b8 & 1) * 8;
            int mode = mb->i4mode[(b8 >> 1) * 8 + (b8 & 1) * 2];
            if (pred8x8l(s, b8, mode, dst, pic->stride_y) < 0) { snprintf(s->d->err, sizeof s->d->err, "Intra8x8 mode %d needs unavailable neighbour", mode);
                return -1; }
            if (cbp & (1 << b8)) recon8x8(s, b8, s->pps->scaling8[0], qp, dst, pic->stride_y);
        }
    } else {                               /* Intra 4x4: predict + residual per block in decode order */
        const uint8_t *list = s->pps->scaling4[0];
        for (int blk = 0; blk < 16; blk++) {
            int bx = blk_x(blk), by = blk_y(blk), r = by * 4 + bx;
            uint8_t *dst = dy + by * 4 * pic->stride_y + bx * 4;
            int mode = mb->i4mode[r];
            /* modes that require unavailable samples are a bitstream error */
            int availA = bx > 0 || intra_usable(s, mb_at(s, s->mb_x - 1, s->mb_y));
            int availB = by > 0 || intra_usable(s, mb_at(s, s->mb_x, s->mb_y - 1));
            if (((mode == 0 || mode == 3 || mode == 7) && !availB) || ((mode == 1 || mode == 8) && !availA) ||
                ((mode == 4 || mode == 5 || mode == 6) && !(availA && availB))) {
                snprintf(s->d->err, sizeof s->d->err, "Intra4x4 mode %d needs unavailable neighbour", mode); return -1;
            }
            pred4x4(s, blk, mode, dst, pic->stride_y);
            if (mb->total_coeff[r]) {
                int dq[16], res[16];
                scale4x4(s->luma[r], dq, list, qp, 0);
                idct4x4(dq, res);
                add_block(dst, pic->stride_y, res, 4);
            }
        }
    }
    if (mb->is_intra) {
        if (pred_chroma(s, s->chroma_pred_mode, du, pic->stride_c) < 0 || pred_chroma(s, s->chroma_pred_mode, dv, pic->stride_c) < 0) {
            snprintf(s->d->err, sizeof s->d->err, "chroma intra mode needs unavailable neighbour"); return -1;
        }
    }
    recon_chroma_residual(s, cbp);
    return 0;
}

/* 7.3.4 slice_data() */
int orc_decode_slice_data(OrcDec *d, Bits *b) {
    Sl *s = (Sl *)calloc(1, sizeof(Sl));
    if (!s) return -1;
    s->d = d; s->b = b; s->pic = d->cur; s->sps = d->asps; s->pps = d->apps; s->sh = &d->sh;
    s->qp = d->sh.qp;
    int n_mbs = d->mb_w * d->mb_h, addr = d->sh.first_mb, rc = 0;
    s->cabac_on = s->pps->entropy_coding_mode;
    d->stats[s->cabac_on ? ORC_ST_CABAC_SLICES : ORC_ST_CAVLC_SLICES]++;
    if (s->cabac_on && d->sh.slice_type != SLICE_I && d->sh.cabac_init_idc <= 2) d->stats[ORC_ST_IDC0 + d->sh.cabac_init_idc]++;
    if (s->cabac_on) {
        while (!bits_aligned(b)) if (!bits_u1(b)) { snprintf(d->err, sizeof d->err, "cabac_alignment_one_bit != 1"); free(s); return -1; }
        orc_cabac_init_contexts(&s->c, d->sh.slice_type == SLICE_I, d->sh.cabac_init_idc, d->sh.qp);
        if (d->sh.cabac_init_idc > 2 || orc_cabac_init_engine(&s->c, b) < 0) { snprintf(d->err, sizeof d->err, "bad CABAC slice start"); free(s); return -1; }
        for (;;) {
            if (addr >= n_mbs) { snprintf(d->err, sizeof d->err, "slice runs past the end of the picture"); rc = -1; break; }
            s->mb_addr = addr; s->mb_x = addr % d->mb_w; s->mb_y = addr / d->mb_w; s->mb = &s->pic->mbs[addr];
            int skip = d->sh.slice_type != SLICE_I ? orc_cabac_mb_skip_flag(s) : 0;
            if ((skip ? decode_skip_mb(s) : decode_mb(s)) < 0) { if (!d->err[0]) snprintf(d->err, sizeof d->err, "macroblock %d decode error", addr); rc = -1;
                break; }
            addr++; d->cur_mb_count++;
            if (b->err) { snprintf(d->err, sizeof d->err, "slice data truncated"); rc = -1; break; }
            if (orc_cabac_terminate(&s->c)) {                  /* end_of_slice_flag */
                /* 9.3.3.2.2.3: the last bit the engine read is rbsp_stop_one_bit, so the slice data ended EXACTLY where the RBSP's trailing bits
                 * begin: the bit before the read position is 1 and nothing but alignment zeros follows.  One wrong context-table entry,
                 * binarisation or ctxIdxInc desynchronises the engine long before this point (hard part 1): counted per slice. */
                size_t p = b->pos; int exact = p > 0 && p <= b->nbits && ((b->p[(p - 1) >> 3] >> (7 - ((p - 1) & 7))) & 1);
                for (size_t q = p; exact && q < b->nbits; q++) if ((b->p[q >> 3] >> (7 - (q & 7))) & 1) exact = 0;
                if (exact) d->stats[ORC_ST_EXACT_END]++;
                break;
            }
        }
        free(s);
        return rc;
    }
    int more = 1;
    while (more) {
        if (d->sh.slice_type != SLICE_I) {
            unsigned run = bits_ue(b);
            if (b->err || run > (unsigned)(n_mbs - addr)) { snprintf(d->err, sizeof d->err, "bad mb_skip_run"); rc = -1; break; }
            for (unsigned i = 0; i < run; i++) {
                s->mb_addr = addr; s->mb_x = addr % d->mb_w; s->mb_y = addr / d->mb_w; s->mb = &s->pic->mbs[addr];
                if (decode_skip_mb(s) < 0) { rc = -1; break; }
                addr++; d->cur_mb_count++;
            }
            if (rc < 0) break;
            if (run > 0) more = bits_more_rbsp(b);
            if (!more) break;
        }
        if (addr >= n_mbs) { snprintf(d->err, sizeof d->err, "slice runs past the end of the picture"); rc = -1; break; }
        s->mb_addr = addr; s->mb_x = addr % d->mb_w; s->mb_y = addr / d->mb_w; s->mb = &s->pic->mbs[addr];
        if (decode_mb(s) < 0) { if (!d->err[0]) snprintf(d->err, sizeof d->err, "macroblock %d decode error", addr); rc = -1; break; }
        addr++; d->cur_mb_count++;
        more = bits_more_rbsp(b);
    }
    if (rc == 0 && !b->err) d->stats[ORC_ST_EXACT_END]++;    /* CAVLC: the loop ends exactly when only rbsp_trailing_bits remain (7.3.4) */
    if (rc == 0 && b->err) { snprintf(d->err, sizeof d->err, "slice data truncated"); rc = -1; }
    free(s);
    return rc;
}
