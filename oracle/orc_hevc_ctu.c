/*
 * oracle/orc_hevc_ctu.c -- CPU ORACLE (test infrastructure only): HEVC slice segment data.
 * CABAC parsing (ITU-T H.265 9.3), coding tree / coding unit / prediction unit / transform tree syntax (7.3.8),
 * intra prediction (8.4), inter prediction incl. merge, AMVP and temporal candidates (8.5), scaling + transforms (8.6).
 * Restates the slice-level part of cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:33-41); spec-literal and
 * bit-serial on purpose, it shares no code with the product's parser (jmcodec_amd/csrc/hevc_*.cpp).
 */
#include "orc_hevc_internal.h"

typedef struct {
    OrchDec *d; HSlice *sh; const HSps *sps; const HPps *pps; int slice_idx;
    Bits b; uint32_t range, offset;
    uint8_t st[ORCH_N_CTX], mps[ORCH_N_CTX];
    int ctb_addr_rs, ctb_addr_ts;
    int qp_y, qp_y_prev, is_dqp_coded, dqp, qg_x, qg_y, first_qg;     /* 8.6.1 */
    int tq_bypass;                /* cu_transquant_bypass_flag of the current CU */
    int cu_intra, cu_x, cu_y, cu_log2, part_mode, ipm_c, cu_skip;
    int max_tr_depth, intra_split, last_merge, qg_started;
    int16_t coef[32 * 32];
    int err;
} Sx;

static void dg_trace(int v);
static void dg(OrchDec *d, int v) {
    if (!d->digest_on) return;
    dg_trace(v);
    for (int i = 0; i < 4; i++) { d->digest ^= (uint8_t)((uint32_t)v >> (8 * i)); d->digest *= 0x100000001b3ULL; }
}

/* ------------------------------------------ 9.3 CABAC ------------------------------------------ */
static void cabac_init_ctx(Sx *s) {                                     /* 9.3.2.2 */
    int t = s->sh->type == H_SLICE_I ? 0 : (s->sh->type == H_SLICE_P ? (s->sh->cabac_init_flag ? 2 : 1) : (s->sh->cabac_init_flag ? 1 : 2));
    int qp = h_clip3(0, 51, s->sh->slice_qp);
    for (int i = 0; i < ORCH_N_CTX; i++) {
        int v = orch_ctx_init[t][i], m = (v >> 4) * 5 - 45, n = ((v & 15) << 3) - 16;
        int pre = h_clip3(1, 126, ((m * qp) >> 4) + n);
        s->mps[i] = pre <= 63 ? 0 : 1;
        s->st[i] = (uint8_t)(s->mps[i] ? pre - 64 : 63 - pre);
    }
}
static void cabac_init_engine(Sx *s) { s->range = 510; s->offset = bits_u(&s->b, 9); }      /* 9.3.2.5 */
static FILE *g_trace; static int g_trace_init;
#define TRACE(...) do { if (!g_trace_init) { g_trace_init = 1; if (getenv("ORCH_TRACE")) g_trace = fopen(getenv("ORCH_TRACE"), "w"); } \
        if (g_trace) fprintf(g_trace, __VA_ARGS__); } while (0)
static void dg_trace(int v) { TRACE("D %d\n", v); }
static int ae_(Sx *s, int ctx);
static int ae(Sx *s, int ctx) { int b = ae_(s, ctx); TRACE("c%d %d\n", ctx, b); return b; }
static int ae_(Sx *s, int ctx) {                                         /* 9.3.4.3.2 */
    uint32_t lps = orch_range_lps[s->st[ctx]][(s->range >> 6) & 3];
    int bin;
    s->range -= lps;
    if (s->offset >= s->range) {
        bin = !s->mps[ctx]; s->offset -= s->range; s->range = lps;
        if (s->st[ctx] == 0) s->mps[ctx] ^= 1;
        s->st[ctx] = orch_trans_lps[s->st[ctx]];
    } else { bin = s->mps[ctx]; if (s->st[ctx] < 62) s->st[ctx]++; }
    while (s->range < 256) { s->range <<= 1; s->offset = (s->offset << 1) | bits_u1(&s->b); }
    return bin;
}
static int ae_bypass(Sx *s) {                                           /* 9.3.4.3.4 */
    s->offset = (s->offset << 1) | bits_u1(&s->b);
    if (s->offset >= s->range) { s->offset -= s->range; TRACE("b 1\n"); return 1; }
    TRACE("b 0\n");
    return 0;
}
static int ae_bypass_n(Sx *s, int n) { int v = 0; while (n--) v = (v << 1) | ae_bypass(s); return v; }
static int ae_terminate(Sx *s) {                                        /* 9.3.4.3.5 */
    s->range -= 2;
    if (s->offset >= s->range) { TRACE("t 1\n"); return 1; }
    TRACE("t 0\n");
    while (s->range < 256) { s->range <<= 1; s->offset = (s->offset << 1) | bits_u1(&s->b); }
    return 0;
}
/* after a terminating bin equal to 1 the arithmetic decoder has read exactly through the encoder's flush (9.3.2.5, 9.3.4.3.5):
 * what follows (PCM samples / the next substream) starts at the next byte boundary */
static void cabac_align(Sx *s) { s->b.pos = (s->b.pos + 7) & ~(size_t)7; }

/* ------------------------------------------ availability (6.4.1) ------------------------------------------ */
static int avail_zs(const Sx *s, int xc, int yc, int xn, int yn) {
    const OrchDec *d = s->d;
    if (xn < 0 || yn < 0 || xn >= d->w || yn >= d->h) return 0;
    int sh = s->sps->log2_min_tb;
    if (d->min_tb_zs[(yn >> sh) * d->tb_w + (xn >> sh)] > d->min_tb_zs[(yc >> sh) * d->tb_w + (xc >> sh)]) return 0;
    int cn = (yn >> s->sps->log2_ctb) * d->ctb_w + (xn >> s->sps->log2_ctb);
    if (d->ctb_slice_addr[cn] != s->sh->slice_addr) return 0;          /* different slice (or not decoded yet) */
    if (d->tile_id[d->ctb_rs2ts[cn]] != d->tile_id[s->ctb_addr_ts]) return 0;
    return 1;
}
#define I4(d, x, y) (((y) >> 2) * (d)->w4 + ((x) >> 2))
/* 6.4.2 prediction block availability */
static int avail_pb(const Sx *s, int xcb, int ycb, int ncbs, int xpb, int ypb, int npbw, int npbh, int part_idx, int xn, int yn) {
    int same_cb = xcb <= xn && ycb <= yn && xcb + ncbs > xn && ycb + ncbs > yn, av;
    if (!same_cb) av = avail_zs(s, xpb, ypb, xn, yn);
    else av = !((npbw << 1) == ncbs && (npbh << 1) == ncbs && part_idx == 1 && ycb + npbh <= yn && xcb + npbw > xn);
    if (av && s->d->pred_mode[I4(s->d, xn, yn)] != 1) av = 0;
    return av;
}

/* ------------------------------------------ 8.4.4.2 intra sample prediction ------------------------------------------ */
static void intra_pred(Sx *s, int x0, int y0, int log2, int c, int mode) {
    OrchDec *d = s->d;
    const int n = 1 << log2, sc = c ? 1 : 0;                          /* sc: chroma subsampling shift */
    uint8_t *pl = d->cur->pl[c]; const int stride = d->cur->stride[c];
    int left_[129], top_[129];                                          /* p[-1][-1 .. 2n-1] at index y+1, p[-1 .. 2n-1][-1] at index x+1 */
    uint8_t avl[129], avt[129];
    int *left = left_, *top = top_;
    int xl = x0 << sc, yl = y0 << sc;                                   /* luma location of the block */
    int unit = c ? 2 : 4;                                               /* samples per availability unit (4 luma samples) */
    int any = 0;
    for (int i = 0; i < 2 * n; i += unit) {
        int a = avail_zs(s, xl, yl, xl - 1, yl + (i << sc));
        if (a && s->pps->constrained_intra && d->pred_mode[I4(d, xl - 1, yl + (i << sc))] != 2) a = 0;
        for (int k = 0; k < unit; k++) { avl[i + k + 1] = (uint8_t)a; if (a) left[i + k + 1] = pl[(y0 + i + k) * stride + x0 - 1]; }
        any |= a;
        a = avail_zs(s, xl, yl, xl + (i << sc), yl - 1);
        if (a && s->pps->constrained_intra && d->pred_mode[I4(d, xl + (i << sc), yl - 1)] != 2) a = 0;
        for (int k = 0; k < unit; k++) { avt[i + k + 1] = (uint8_t)a; if (a) top[i + k + 1] = pl[(y0 - 1) * stride + x0 + i + k]; }
        any |= a;
    }
    { int a = avail_zs(s, xl, yl, xl - 1, yl - 1);
      if (a && s->pps->constrained_intra && d->pred_mode[I4(d, xl - 1, yl - 1)] != 2) a = 0;
      avl[0] = avt[0] = (uint8_t)a; if (a) left[0] = top[0] = pl[(y0 - 1) * stride + x0 - 1];
      any |= a; }
    /* 8.4.4.2.2 substitution */
    if (!any) { for (int i = 0; i <= 2 * n; i++) left[i] = top[i] = 128; }
    else {
        if (!avl[2 * n]) {
            int v = -1;
            for (int i = 2 * n - 1; i >= 0 && v < 0; i--) if (avl[i]) v = left[i];
            for (int i = 1; i <= 2 * n && v < 0; i++) if (avt[i]) v = top[i];
            left[2 * n] = v; avl[2 * n] = 1;
        }
        for (int i = 2 * n - 1; i >= 0; i--) if (!avl[i]) { left[i] = left[i + 1]; avl[i] = 1; }
        top[0] = left[0];
        for (int i = 1; i <= 2 * n; i++) if (!avt[i]) top[i] = top[i - 1];
    }
    /* 8.4.4.2.3 filtering of neighbouring samples */
    int fl_[129], ft_[129];
    if (c == 0 && mode != 1 && n != 4) {
        int md = abs(mode - 26) < abs(mode - 10) ? abs(mode - 26) : abs(mode - 10);
        int thr = n == 8 ? 7 : (n == 16 ? 1 : 0);
        if (md > thr) {
            if (s->sps->strong_intra_smoothing && n == 32 && abs(top[0] + top[64] - 2 * top[32]) < 8 && abs(left[0] + left[64] - 2 * left[32]) < 8) {
                d->stats[HST_STRONG_INTRA]++;
                fl_[0] = ft_[0] = top[0];
                for (int i = 0; i < 63; i++) { fl_[i + 1] = ((63 - i) * left[0] + (i + 1) * left[64] + 32) >> 6;
                    ft_[i + 1] = ((63 - i) * top[0] + (i + 1) * top[64] + 32) >> 6; }
                fl_[64] = left[64]; ft_[64] = top[64];
            } else {
                fl_[0] = ft_[0] = (left[1] + 2 * left[0] + top[1] + 2) >> 2;
                for (int i = 1; i < 2 * n; i++) { fl_[i] = (left[i + 1] + 2 * left[i] + left[i - 1] + 2) >> 2;
                    ft_[i] = (top[i + 1] + 2 * top[i] + top[i - 1] + 2) >> 2; }
                fl_[2 * n] = left[2 * n]; ft_[2 * n] = top[2 * n];
            }
            left = fl_; top = ft_;
        }
    }
#define PL(y) left[(y) + 1]
#define PT(x) top[(x) + 1]
    uint8_t *dst = pl + y0 * stride + x0;
    if (mode == 0) {                                                    /* 8.4.4.2.4 planar */
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++)
            dst[y * stride + x] = (uint8_t)(((n - 1 - x) * PL(y) + (x + 1) * PT(n) + (n - 1 - y) * PT(x) + (y + 1) * PL(n) + n) >> (log2 + 1));
    } else if (mode == 1) {                                             /* 8.4.4.2.5 DC */
        int sum = n;
        for (int i = 0; i < n; i++) sum += PL(i) + PT(i);
        int dc = sum >> (log2 + 1);
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) dst[y * stride + x] = (uint8_t)dc;
        if (c == 0 && n < 32) {
            dst[0] = (uint8_t)((PL(0) + 2 * dc + PT(0) + 2) >> 2);
            for (int x = 1; x < n; x++) dst[x] = (uint8_t)((PT(x) + 3 * dc + 2) >> 2);
            for (int y = 1; y < n; y++) dst[y * stride] = (uint8_t)((PL(y) + 3 * dc + 2) >> 2);
        }
    } else {                                                            /* 8.4.4.2.6 angular */
        int ang = orch_intra_angle[mode], inv = orch_inv_angle[mode];
        int ref_[3 * 32 + 4], *ref = ref_ + 32;
        if (mode >= 18) {
            for (int x = 0; x <= n; x++) ref[x] = PT(x - 1);
            if (ang < 0) { int last = (n * ang) >> 5; if (last < -1) for (int x = last; x <= -1; x++) ref[x] = PL(-1 + ((x * inv + 128) >> 8)); }
            else for (int x = n + 1; x <= 2 * n; x++) ref[x] = PT(x - 1);
            for (int y = 0; y < n; y++) {
                int idx = ((y + 1) * ang) >> 5, f = ((y + 1) * ang) & 31;
                for (int x = 0; x < n; x++) dst[y * stride + x] =
                    (uint8_t)(f ? ((32 - f) * ref[x + idx + 1] + f * ref[x + idx + 2] + 16) >> 5 : ref[x + idx + 1]);
            }
            if (mode == 26 && c == 0 && n < 32) for (int y = 0; y < n; y++) dst[y * stride] = (uint8_t)h_clip1(PT(0) + ((PL(y) - PL(-1)) >> 1));
        } else {
            for (int x = 0; x <= n; x++) ref[x] = PL(x - 1);
            if (ang < 0) { int last = (n * ang) >> 5; if (last < -1) for (int x = last; x <= -1; x++) ref[x] = PT(-1 + ((x * inv + 128) >> 8)); }
            else for (int x = n + 1; x <= 2 * n; x++) ref[x] = PL(x - 1);
            for (int x = 0; x < n; x++) {
                int idx = ((x + 1) * ang) >> 5, f = ((x + 1) * ang) & 31;
                for (int y = 0; y < n; y++) dst[y * stride + x] =
                    (uint8_t)(f ? ((32 - f) * ref[y + idx + 1] + f * ref[y + idx + 2] + 16) >> 5 : ref[y + idx + 1]);
            }
            if (mode == 10 && c == 0 && n < 32) for (int x = 0; x < n; x++) dst[x] = (uint8_t)h_clip1(PL(0) + ((PT(x) - PT(-1)) >> 1));
        }
    }
#undef PL
#undef PT
}

/* ------------------------------------------ 8.6 scaling, transformation, reconstruction ------------------------------------------ */
static void transform_1d(const int *in, int *out, int n, int dst_type) {           /* 8.6.4.2: out[i] = sum_j M[j][i] * in[j] */
    if (dst_type) { for (int i = 0; i < 4; i++) { int v = 0; for (int j = 0; j < 4; j++) v += orch_dst[j][i] * in[j]; out[i] = v; } return; }
    int step = 32 / n;
    for (int i = 0; i < n; i++) { int v = 0; for (int j = 0; j < n; j++) v += orch_trans[j * step][i] * in[j]; out[i] = v; }
}
/* levels in s->coef (raster, y * n + x) -> residual added to the picture at (x0, y0) of plane c */
static void residual_add(Sx *s, int x0, int y0, int log2, int c, int tskip, int qp) {
    OrchDec *d = s->d;
    const int n = 1 << log2;
    int r[32 * 32];
    if (s->tq_bypass) { for (int i = 0; i < n * n; i++) r[i] = s->coef[i]; }
    else {
        /* 8.6.4.1 scaling */
        const ScalingFactors *sf = s->pps->scaling_present ? &s->pps->sf : &s->sps->sf;
        int mat = (s->cu_intra ? 0 : 3) + c;
        const uint8_t *m = log2 == 2 ? sf->f4[mat] : log2 == 3 ? sf->f8[mat] : log2 == 4 ? sf->f16[mat] : sf->f32[s->cu_intra ? 0 : 1];
        int flat = !s->sps->scaling_list_enabled || (tskip && n > 4);
        int bd_shift = 8 + log2 - 5, ls = orch_level_scale[qp % 6] << (qp / 6);
        for (int i = 0; i < n * n; i++)
            r[i] = s->coef[i] ? h_clip3(-32768, 32767, (int)(((int64_t)s->coef[i] * (flat ? 16 : m[i]) * ls + (1 << (bd_shift - 1))) >> bd_shift)) : 0;
        if (tskip) { for (int i = 0; i < n * n; i++) r[i] = ((r[i] << 7) + (1 << 11)) >> 12; }
        else {
            int dst_type = s->cu_intra && c == 0 && n == 4;
            int col[32], o[32], g[32 * 32];
            for (int x = 0; x < n; x++) {                               /* columns first */
                for (int y = 0; y < n; y++) col[y] = r[y * n + x];
                transform_1d(col, o, n, dst_type);
                for (int y = 0; y < n; y++) g[y * n + x] = h_clip3(-32768, 32767, (o[y] + 64) >> 7);
            }
            for (int y = 0; y < n; y++) {
                transform_1d(g + y * n, o, n, dst_type);
                for (int x = 0; x < n; x++) r[y * n + x] = (o[x] + (1 << 11)) >> 12;
            }
        }
    }
    uint8_t *dst = d->cur->pl[c] + y0 * d->cur->stride[c] + x0;
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) dst[y * d->cur->stride[c] + x] = (uint8_t)h_clip1(dst[y * d->cur->stride[c] + x] + r[y * n + x]);
}

/* ------------------------------------------ 7.3.8.11 residual_coding ------------------------------------------ */
static void scan_pos(int scan_idx, int log2blk, int i, int *x, int *y) {            /* 6.5.3-6.5.5 for blocks of 1 << log2blk (<= 8) */
    int n = 1 << log2blk;
    if (scan_idx == 1) { *x = i % n; *y = i / n; return; }
    if (scan_idx == 2) { *x = i / n; *y = i % n; return; }
    int k = 0, xx = 0, yy = 0;
    for (;;) {
        while (yy >= 0) { if (xx < n && yy < n) { if (k == i) { *x = xx; *y = yy; return; } k++; } yy--; xx++; }
        yy = xx; xx = 0;
    }
}
static int parse_last_prefix(Sx *s, int base, int log2, int c) {
    int cmax = (log2 << 1) - 1, off, shift, v = 0;
    if (c == 0) { off = 3 * (log2 - 2) + ((log2 - 1) >> 2); shift = (log2 + 1) >> 2; } else { off = 15; shift = log2 - 2; }
    while (v < cmax && ae(s, base + off + (v >> shift))) v++;
    return v;
}
static int residual_coding(Sx *s, int x0, int y0, int log2, int c, int *tskip_out) {
    OrchDec *d = s->d;
    const int n = 1 << log2;
    int tskip = 0;
    memset(s->coef, 0, sizeof(int16_t) * (size_t)(n * n));
    if (s->pps->transform_skip && !s->tq_bypass && log2 == 2) tskip = ae(s, ORCH_CTX_TSKIP + (c ? 1 : 0));
    int lx = parse_last_prefix(s, ORCH_CTX_LAST_X, log2, c), ly = parse_last_prefix(s, ORCH_CTX_LAST_Y, log2, c);
    if (lx > 3) { int nb = (lx >> 1) - 1; lx = (1 << nb) * (2 + (lx & 1)) + ae_bypass_n(s, nb); }
    if (ly > 3) { int nb = (ly >> 1) - 1; ly = (1 << nb) * (2 + (ly & 1)) + ae_bypass_n(s, nb); }
    int scan_idx = 0;
    if (s->cu_intra && (log2 == 2 || (log2 == 3 && c == 0))) {
        int pm = c == 0 ? d->ipm[I4(d, x0, y0)] : s->ipm_c;
        if (pm >= 6 && pm <= 14) scan_idx = 2; else if (pm >= 22 && pm <= 30) scan_idx = 1;
    }
    if (scan_idx == 2) { int t = lx; lx = ly; ly = t; }
    if (lx >= n || ly >= n) { s->err = 1; return -1; }
    int nsb_log2 = log2 - 2, last_sb = (1 << (2 * nsb_log2)) - 1, last_pos = 16, xs, ys, xp, yp;
    do {
        if (last_pos == 0) { last_pos = 16; last_sb--; }
        last_pos--;
        if (last_sb < 0) { s->err = 1; return -1; }
        scan_pos(scan_idx, nsb_log2, last_sb, &xs, &ys); scan_pos(scan_idx, 2, last_pos, &xp, &yp);
    } while ((xs << 2) + xp != lx || (ys << 2) + yp != ly);
    uint8_t csbf[8][8]; memset(csbf, 0, sizeof csbf);
    int greater1_ctx = 1, first_sb = 1, nsb = 1 << nsb_log2;
    for (int i = last_sb; i >= 0; i--) {
        scan_pos(scan_idx, nsb_log2, i, &xs, &ys);
        int infer_dc = 0;
        if (i < last_sb && i > 0) {
            int ctx = 0;
            if (xs < nsb - 1) ctx |= csbf[ys][xs + 1];
            if (ys < nsb - 1) ctx |= csbf[ys + 1][xs];
            csbf[ys][xs] = (uint8_t)ae(s, ORCH_CTX_CSBF + ctx + (c ? 2 : 0));
            infer_dc = 1;
        } else csbf[ys][xs] = 1;
        uint8_t sig[16]; memset(sig, 0, sizeof sig);
        int start = 15;
        if (i == last_sb) { start = last_pos - 1; sig[last_pos] = 1; }
        if (csbf[ys][xs]) {
            int prev = 0;
            if (xs < nsb - 1) prev |= csbf[ys][xs + 1];
            if (ys < nsb - 1) prev |= csbf[ys + 1][xs] << 1;
            for (int k = start; k >= 0; k--) {
                scan_pos(scan_idx, 2, k, &xp, &yp);
                if (k > 0 || !infer_dc) {
                    int sc;                                             /* 9.3.4.2.5 */
                    int xc = (xs << 2) + xp, yc = (ys << 2) + yp;
                    if (log2 == 2) { static const uint8_t map[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8}; sc = map[(yc << 2) + xc]; }
                    else if (xc + yc == 0) sc = 0;
                    else {
                        if (prev == 0) sc = (xp + yp == 0) ? 2 : (xp + yp < 3) ? 1 : 0;
                        else if (prev == 1) sc = yp == 0 ? 2 : (yp == 1 ? 1 : 0);
                        else if (prev == 2) sc = xp == 0 ? 2 : (xp == 1 ? 1 : 0);
                        else sc = 2;
                        if (c == 0) { if (xs + ys > 0) sc += 3; sc += log2 == 3 ? (scan_idx == 0 ? 9 : 15) : 21; }
                        else sc += log2 == 3 ? 9 : 12;
                    }
                    sig[k] = (uint8_t)ae(s, ORCH_CTX_SIG + (c == 0 ? sc : 27 + sc));
                    if (sig[k]) infer_dc = 0;
                } else sig[k] = 1;                                      /* k == 0 with inferSbDcSigCoeffFlag */
            }
        }
        int pos[16], npos = 0;
        for (int k = 15; k >= 0; k--) if (sig[k]) pos[npos++] = k;
        if (!npos) continue;
        /* greater1 / greater2 flags (9.3.4.2.6, 9.3.4.2.7) */
        int ctx_set = (i == 0 || c > 0) ? 0 : 2;
        if (!first_sb && greater1_ctx == 0) ctx_set++;
        first_sb = 0; greater1_ctx = 1;
        int g1[16] = {0}, g2 = 0, last_g1 = -1;
        for (int m = 0; m < npos && m < 8; m++) {
            g1[m] = ae(s, ORCH_CTX_G1 + (ctx_set << 2) + greater1_ctx + (c ? 16 : 0));
            if (g1[m]) { greater1_ctx = 0; if (last_g1 < 0) last_g1 = m; }
            else if (greater1_ctx > 0 && greater1_ctx < 3) greater1_ctx++;
        }
        if (last_g1 >= 0) g2 = ae(s, ORCH_CTX_G2 + ctx_set + (c ? 4 : 0));
        int sign_hidden = !s->tq_bypass && (pos[0] - pos[npos - 1] > 3);
        int hide = s->pps->sign_hiding && sign_hidden;
        int nsign = npos - (hide ? 1 : 0);
        uint32_t signs = (uint32_t)ae_bypass_n(s, nsign) << (16 - nsign);
        int rice = 0, sum = 0;
        for (int m = 0; m < npos; m++) {
            int base = 1 + g1[m] + (m == last_g1 ? g2 : 0);
            int thresh = m < 8 ? (m == last_g1 ? 3 : 2) : 1;
            int lev = base;
            if (base == thresh) {                                       /* coeff_abs_level_remaining (9.3.3.11) */
                int q = 0;
                while (q < 32 && ae_bypass(s)) q++;
                if (q >= 32) { s->err = 1; return -1; }
                int rem;
                if (q < 4) rem = (q << rice) + ae_bypass_n(s, rice);
                else { int nb = q - 3 + rice; if (nb > 30) { s->err = 1; return -1; } rem = (((1 << (q - 3)) + 3 - 1) << rice) + ae_bypass_n(s, nb); }
                lev = base + rem;
                if (lev > 3 * (1 << rice)) rice = rice < 4 ? rice + 1 : 4;
            }
            sum += lev;
            int neg;
            if (hide && m == npos - 1) neg = sum & 1;
            else { neg = (signs >> 15) & 1; signs <<= 1; }
            scan_pos(scan_idx, 2, pos[m], &xp, &yp);
            int v = neg ? -lev : lev;
            s->coef[((ys << 2) + yp) * n + (xs << 2) + xp] = (int16_t)h_clip3(-32768, 32767, v);
        }
        if (hide) d->stats[HST_SDH]++;
    }
    if (d->digest_on) {
        dg(d, 0x7000 | (c << 8) | (log2 << 4) | tskip); dg(d, x0); dg(d, y0);
        for (int k = 0; k < n * n; k++) if (s->coef[k]) { dg(d, k); dg(d, s->coef[k]); }
    }
    if (tskip) d->stats[HST_TSKIP]++;
    *tskip_out = tskip;
    return 0;
}

/* ------------------------------------------ 8.5 inter prediction ------------------------------------------ */
typedef struct { int16_t mv[2][2]; int8_t ref[2]; uint8_t pf; } Cand;
static Cand cand_of(const OrchDec *d, int x, int y) { const HMotion *m = &d->mot[I4(d, x, y)]; Cand c; memcpy(c.mv, m->mv, sizeof c.mv);
    c.ref[0] = m->ref_idx[0]; c.ref[1] = m->ref_idx[1]; c.pf = m->pred_flag; return c; }
static int cand_same(const Cand *a, const Cand *b) {
    if (a->pf != b->pf) return 0;
    for (int l = 0; l < 2; l++) if (a->pf & (1 << l)) { if (a->ref[l] != b->ref[l] || a->mv[l][0] != b->mv[l][0] || a->mv[l][1] != b->mv[l][1]) return 0; }
    return 1;
}
static int mv_scale(int mv, int td, int tb) {                           /* (8-179 .. 8-183) */
    td = h_clip3(-128, 127, td); tb = h_clip3(-128, 127, tb);
    int tx = (16384 + (abs(td) >> 1)) / td;
    int f = h_clip3(-4096, 4095, (tb * tx + 32) >> 6);
    int p = f * mv;
    return h_clip3(-32768, 32767, (p < 0 ? -1 : 1) * ((abs(p) + 127) >> 8));
}
/* 8.5.3.2.8 / 8.5.3.2.9 temporal luma motion vector prediction for list X and refIdx */
static int temporal_mv(Sx *s, int xpb, int ypb, int npbw, int npbh, int X, int ref_idx, int16_t mv_out[2]) {
    OrchDec *d = s->d; HSlice *sh = s->sh;
    if (!sh->temporal_mvp) return 0;
    int cl = sh->type == H_SLICE_B && !sh->collocated_from_l0 ? 1 : 0;
    int ci = sh->ref_dpb[cl][sh->collocated_ref_idx];
    if (ci < 0) return 0;
    const HPic *col = &d->dpb[ci];
    if (!col->col_mv) return 0;
    for (int pass = 0; pass < 2; pass++) {
        int xc, yc;
        if (pass == 0) {
            xc = xpb + npbw; yc = ypb + npbh;
            if ((ypb >> s->sps->log2_ctb) != (yc >> s->sps->log2_ctb) || yc >= d->h || xc >= d->w) continue;
        } else { xc = xpb + (npbw >> 1); yc = ypb + (npbh >> 1); }
        int ce = (yc >> 4) * col->col_w + (xc >> 4);
        if (col->col_intra[ce]) continue;
        const HMotion *cm = &col->col_mv[ce];
        int lc;                                                         /* which list of the collocated block */
        if (!(cm->pred_flag & 1)) lc = 1;
        else if (!(cm->pred_flag & 2)) lc = 0;
        else {
            int no_backward = 1; /* NoBackwardPredFlag: DiffPicOrderCnt(aPic, currPic) <= 0 for every reference picture */
            for (int l = 0; l < 2; l++) for (int i = 0; i < sh->n_ref[l]; i++) if (sh->ref_poc[l][i] > d->cur->poc) no_backward = 0;
            lc = no_backward ? X : sh->collocated_from_l0;
        }
        int col_ref_poc = col->col_ref_poc[ce * 2 + lc], col_lt = (col->col_ref_lt[ce] >> lc) & 1;
        if (col_lt != sh->ref_is_lt[X][ref_idx]) continue;
        int col_diff = col->poc - col_ref_poc, cur_diff = d->cur->poc - sh->ref_poc[X][ref_idx];
        if (sh->ref_is_lt[X][ref_idx] || col_diff == cur_diff || col_diff == 0) { mv_out[0] = cm->mv[lc][0]; mv_out[1] = cm->mv[lc][1]; }
        else { mv_out[0] = (int16_t)mv_scale(cm->mv[lc][0], col_diff, cur_diff); mv_out[1] = (int16_t)mv_scale(cm->mv[lc][1], col_diff, cur_diff); }
        d->stats[HST_TMVP]++;
        return 1;
    }
    return 0;
}
/* 8.5.3.2.2 merge mode */
static Cand merge_cand(Sx *s, int xcb, int ycb, int ncbs, int xpb, int ypb, int npbw, int npbh, int part_idx, int merge_idx) {
    OrchDec *d = s->d; HSlice *sh = s->sh;
    int ow = npbw, oh = npbh;
    int pml = s->pps->log2_par_mrg_level, part_mode = s->part_mode;
    if (pml > 2 && ncbs == 8) { xpb = xcb; ypb = ycb; npbw = npbh = ncbs; part_idx = 0; part_mode = H_PART_2Nx2N; }
    Cand list[6]; int n = 0;
    Cand a1, b1, b0, a0, b2; int fa1, fb1, fb0, fa0, fb2, vb1;      /* vb1: availableB1 (8.5.3.2.3), fb1: availableFlagB1 (after the comparison with A1) */
#define SAME_MER(xn, yn) ((xpb >> pml) == ((xn) >> pml) && (ypb >> pml) == ((yn) >> pml))
    { int xn = xpb - 1, yn = ypb + npbh - 1;
      fa1 = !(SAME_MER(xn, yn) || (part_idx == 1 && (part_mode == H_PART_Nx2N || part_mode == H_PART_nLx2N || part_mode == H_PART_nRx2N))) &&
          avail_pb(s, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xn, yn);
      if (fa1) { a1 = cand_of(d, xn, yn); list[n++] = a1; } }
    { int xn = xpb + npbw - 1, yn = ypb - 1;
      fb1 = !(SAME_MER(xn, yn) || (part_idx == 1 && (part_mode == H_PART_2NxN || part_mode == H_PART_2NxnU || part_mode == H_PART_2NxnD))) &&
          avail_pb(s, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xn, yn);
      vb1 = fb1;
      if (fb1) { b1 = cand_of(d, xn, yn); if (fa1 && cand_same(&a1, &b1)) fb1 = 0; else list[n++] = b1; } }
    { int xn = xpb + npbw, yn = ypb - 1;
      fb0 = !SAME_MER(xn, yn) && avail_pb(s, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xn, yn);
      if (fb0) { b0 = cand_of(d, xn, yn); if (vb1 && cand_same(&b1, &b0)) { fb0 = 0; if (!fb1) d->stats[HST_MERGE_VS_PRUNED_B1]++; } else list[n++] = b0; } }
    { int xn = xpb - 1, yn = ypb + npbh;
      fa0 = !SAME_MER(xn, yn) && avail_pb(s, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xn, yn);
      if (fa0) { a0 = cand_of(d, xn, yn); if (fa1 && cand_same(&a1, &a0)) fa0 = 0; else list[n++] = a0; } }
    { int xn = xpb - 1, yn = ypb - 1;
      fb2 = !SAME_MER(xn, yn) && fa0 + fa1 + fb0 + fb1 != 4 && avail_pb(s, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xn, yn);
      if (fb2) { b2 = cand_of(d, xn, yn); if ((fa1 && cand_same(&a1, &b2)) || (vb1 && cand_same(&b1, &b2))) { fb2 = 0;
          if (vb1 && !fb1 && !(fa1 && cand_same(&a1, &b2))) d->stats[HST_MERGE_VS_PRUNED_B1]++; } else list[n++] = b2; } }
#undef SAME_MER
    /* B0 and B2 are compared with B1 whenever availableB1 is TRUE -- also when B1 was dropped as a duplicate of A1 (availableFlagB1 = 0);
     * the count of four candidates uses the flags (8.5.3.2.3) */
    if (n < sh->max_merge_cand && sh->temporal_mvp) {
        Cand t; memset(&t, 0, sizeof t); t.ref[0] = t.ref[1] = -1;
        if (temporal_mv(s, xpb, ypb, npbw, npbh, 0, 0, t.mv[0])) { t.pf |= 1; t.ref[0] = 0; }
        if (sh->type == H_SLICE_B && temporal_mv(s, xpb, ypb, npbw, npbh, 1, 0, t.mv[1])) { t.pf |= 2; t.ref[1] = 0; }
        if (t.pf) list[n++] = t;
    }
    if (n > sh->max_merge_cand) n = sh->max_merge_cand;
    if (sh->type == H_SLICE_B && n > 1 && n < sh->max_merge_cand) {     /* 8.5.3.2.4 combined bi-predictive candidates */
        static const uint8_t i0[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, i1[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
        int n_orig = n;
        for (int k = 0; k < n_orig * (n_orig - 1) && n < sh->max_merge_cand; k++) {
            const Cand *c0 = &list[i0[k]], *c1 = &list[i1[k]];
            if ((c0->pf & 1) && (c1->pf & 2) && (sh->ref_poc[0][c0->ref[0]] != sh->ref_poc[1][c1->ref[1]] || c0->mv[0][0] != c1->mv[1][0] ||
                c0->mv[0][1] != c1->mv[1][1])) {
                Cand t; t.pf = 3; t.ref[0] = c0->ref[0]; t.ref[1] = c1->ref[1]; t.mv[0][0] = c0->mv[0][0]; t.mv[0][1] = c0->mv[0][1]; t.mv[1][0] = c1->mv[1][0];
                t.mv[1][1] = c1->mv[1][1];
                list[n++] = t;
            }
        }
    }
    { int num_ref = sh->type == H_SLICE_P ? sh->n_ref[0] : (sh->n_ref[0] < sh->n_ref[1] ? sh->n_ref[0] : sh->n_ref[1]);     /* 8.5.3.2.5 zero candidates */
      for (int z = 0; n < sh->max_merge_cand; z++) {
          Cand t; memset(&t, 0, sizeof t);
          t.pf = sh->type == H_SLICE_P ? 1 : 3; t.ref[0] = (int8_t)(z < num_ref ? z : 0);
          t.ref[1] = (int8_t)(sh->type == H_SLICE_P ? -1 : (z < num_ref ? z : 0));
          list[n++] = t;
      } }
    Cand r = list[merge_idx];
    if (r.pf == 3 && ow + oh == 12) { r.pf = 1; r.ref[1] = -1; r.mv[1][0] = r.mv[1][1] = 0; }
    if (!(r.pf & 1)) { r.ref[0] = -1; r.mv[0][0] = r.mv[0][1] = 0; }
    if (!(r.pf & 2)) { r.ref[1] = -1; r.mv[1][0] = r.mv[1][1] = 0; }
    return r;
}
/* 8.5.3.2.6 / 8.5.3.2.7 luma motion vector prediction */
static void amvp(Sx *s, int xcb, int ycb, int ncbs, int xpb, int ypb, int npbw, int npbh, int part_idx, int X, int ref_idx, int mvp_flag, int16_t out[2]) {
    OrchDec *d = s->d; HSlice *sh = s->sh;
    const int Y = !X, tpoc = sh->ref_poc[X][ref_idx], tlt = sh->ref_is_lt[X][ref_idx];
    int xa[2] = {xpb - 1, xpb - 1}, ya[2] = {ypb + npbh, ypb + npbh - 1};
    int xb[3] = {xpb + npbw, xpb + npbw - 1, xpb - 1}, yb[3] = {ypb - 1, ypb - 1, ypb - 1};
    int ava[2], avb[3], fa = 0, fb = 0; int16_t mva[2] = {0, 0}, mvb[2] = {0, 0};
    for (int k = 0; k < 2; k++) ava[k] = avail_pb(s, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xa[k], ya[k]);
    for (int k = 0; k < 3; k++) avb[k] = avail_pb(s, xcb, ycb, ncbs, xpb, ypb, npbw, npbh, part_idx, xb[k], yb[k]);
    int is_scaled = ava[0] || ava[1];
    for (int k = 0; k < 2 && !fa; k++) if (ava[k]) {
        const HMotion *m = &d->mot[I4(d, xa[k], ya[k])];
        const HSlice *ns = &d->slices[d->slice_of4[I4(d, xa[k], ya[k])]];
        if ((m->pred_flag >> X & 1) && ns->ref_poc[X][m->ref_idx[X]] == tpoc) { fa = 1; mva[0] = m->mv[X][0]; mva[1] = m->mv[X][1]; }
        else if ((m->pred_flag >> Y & 1) && ns->ref_poc[Y][m->ref_idx[Y]] == tpoc) { fa = 1; mva[0] = m->mv[Y][0]; mva[1] = m->mv[Y][1]; }
    }
    for (int k = 0; k < 2 && !fa; k++) if (ava[k]) {
        const HMotion *m = &d->mot[I4(d, xa[k], ya[k])];
        const HSlice *ns = &d->slices[d->slice_of4[I4(d, xa[k], ya[k])]];
        int l = -1;
        if ((m->pred_flag >> X & 1) && ns->ref_is_lt[X][m->ref_idx[X]] == tlt) l = X;
        else if ((m->pred_flag >> Y & 1) && ns->ref_is_lt[Y][m->ref_idx[Y]] == tlt) l = Y;
        if (l >= 0) {
            fa = 1; mva[0] = m->mv[l][0]; mva[1] = m->mv[l][1];
            int rp = ns->ref_poc[l][m->ref_idx[l]];
            if (!ns->ref_is_lt[l][m->ref_idx[l]] && !tlt) { int td = d->cur->poc - rp, tb = d->cur->poc - tpoc; if (td != tb && td != 0) {
                mva[0] = (int16_t)mv_scale(mva[0], td, tb); mva[1] = (int16_t)mv_scale(mva[1], td, tb); } }
        }
    }
    for (int k = 0; k < 3 && !fb; k++) if (avb[k]) {
        const HMotion *m = &d->mot[I4(d, xb[k], yb[k])];
        const HSlice *ns = &d->slices[d->slice_of4[I4(d, xb[k], yb[k])]];
        if ((m->pred_flag >> X & 1) && ns->ref_poc[X][m->ref_idx[X]] == tpoc) { fb = 1; mvb[0] = m->mv[X][0]; mvb[1] = m->mv[X][1]; }
        else if ((m->pred_flag >> Y & 1) && ns->ref_poc[Y][m->ref_idx[Y]] == tpoc) { fb = 1; mvb[0] = m->mv[Y][0]; mvb[1] = m->mv[Y][1]; }
    }
    if (!is_scaled && fb) { fa = 1; mva[0] = mvb[0]; mva[1] = mvb[1]; }
    if (!is_scaled) {
        fb = 0;
        for (int k = 0; k < 3 && !fb; k++) if (avb[k]) {
            const HMotion *m = &d->mot[I4(d, xb[k], yb[k])];
            const HSlice *ns = &d->slices[d->slice_of4[I4(d, xb[k], yb[k])]];
            int l = -1;
            if ((m->pred_flag >> X & 1) && ns->ref_is_lt[X][m->ref_idx[X]] == tlt) l = X;
            else if ((m->pred_flag >> Y & 1) && ns->ref_is_lt[Y][m->ref_idx[Y]] == tlt) l = Y;
            if (l >= 0) {
                fb = 1; mvb[0] = m->mv[l][0]; mvb[1] = m->mv[l][1];
                int rp = ns->ref_poc[l][m->ref_idx[l]];
                if (!ns->ref_is_lt[l][m->ref_idx[l]] && !tlt) { int td = d->cur->poc - rp, tb = d->cur->poc - tpoc; if (td != tb && td != 0) {
                    mvb[0] = (int16_t)mv_scale(mvb[0], td, tb); mvb[1] = (int16_t)mv_scale(mvb[1], td, tb); } }
            }
        }
    }
    int16_t list[3][2]; int n = 0;
    if (fa) { list[n][0] = mva[0]; list[n][1] = mva[1]; n++; }
    if (fb && !(fa && mva[0] == mvb[0] && mva[1] == mvb[1])) { list[n][0] = mvb[0]; list[n][1] = mvb[1]; n++; }
    if (n < 2) { int16_t t[2]; if (temporal_mv(s, xpb, ypb, npbw, npbh, X, ref_idx, t)) { list[n][0] = t[0]; list[n][1] = t[1]; n++; } }
    while (n < 2) { list[n][0] = list[n][1] = 0; n++; }
    out[0] = list[mvp_flag][0]; out[1] = list[mvp_flag][1];
}

/* 8.5.3.3.3 fractional sample interpolation -> 14-bit intermediate predSamples */
static void mc_block(const HPic *ref, int c, int pw, int ph, int xb, int yb, int bw, int bh, int mvx, int mvy, int16_t *out) {
    const uint8_t *pl = ref->pl[c]; const int stride = ref->stride[c];
    if (c == 0) {
        int xf = mvx & 3, yf = mvy & 3, xi = xb + (mvx >> 2), yi = yb + (mvy >> 2);
        for (int y = 0; y < bh; y++) for (int x = 0; x < bw; x++) {
            int tmp[8];
            for (int j = 0; j < 8; j++) {
                int yy = h_clip3(0, ph - 1, yi + y + j - 3), v = 0;
                if (xf == 0) v = pl[yy * stride + h_clip3(0, pw - 1, xi + x)];
                else for (int i = 0; i < 8; i++) v += orch_luma_filter[xf][i] * pl[yy * stride + h_clip3(0, pw - 1, xi + x + i - 3)];
                tmp[j] = v;
            }
            int v;
            if (xf == 0 && yf == 0) v = tmp[3] << 6;
            else if (yf == 0) v = tmp[3];
            else if (xf == 0) { v = 0; for (int j = 0; j < 8; j++) v += orch_luma_filter[yf][j] * tmp[j]; }
            else { v = 0; for (int j = 0; j < 8; j++) v += orch_luma_filter[yf][j] * tmp[j]; v >>= 6; }
            out[y * bw + x] = (int16_t)v;
        }
    } else {
        int xf = mvx & 7, yf = mvy & 7, xi = xb + (mvx >> 3), yi = yb + (mvy >> 3);
        for (int y = 0; y < bh; y++) for (int x = 0; x < bw; x++) {
            int tmp[4];
            for (int j = 0; j < 4; j++) {
                int yy = h_clip3(0, ph - 1, yi + y + j - 1), v = 0;
                if (xf == 0) v = pl[yy * stride + h_clip3(0, pw - 1, xi + x)];
                else for (int i = 0; i < 4; i++) v += orch_chroma_filter[xf][i] * pl[yy * stride + h_clip3(0, pw - 1, xi + x + i - 1)];
                tmp[j] = v;
            }
            int v;
            if (xf == 0 && yf == 0) v = tmp[1] << 6;
            else if (yf == 0) v = tmp[1];
            else if (xf == 0) { v = 0; for (int j = 0; j < 4; j++) v += orch_chroma_filter[yf][j] * tmp[j]; }
            else { v = 0; for (int j = 0; j < 4; j++) v += orch_chroma_filter[yf][j] * tmp[j]; v >>= 6; }
            out[y * bw + x] = (int16_t)v;
        }
    }
}
/* 8.5.3.3 decoding process for inter sample prediction of one prediction block */
static int inter_pred(Sx *s, int xpb, int ypb, int npbw, int npbh, const Cand *m) {
    OrchDec *d = s->d; HSlice *sh = s->sh;
    static __thread int16_t p[2][64 * 64];
    int wp = (sh->type == H_SLICE_P && s->pps->weighted_pred) || (sh->type == H_SLICE_B && s->pps->weighted_bipred);
    for (int c = 0; c < 3; c++) {
        int sc = c ? 1 : 0, bw = npbw >> sc, bh = npbh >> sc, xb = xpb >> sc, yb = ypb >> sc, pw = d->w >> sc, ph = d->h >> sc;
        for (int l = 0; l < 2; l++) if (m->pf >> l & 1) {
            int di = sh->ref_dpb[l][m->ref[l]];
            if (di < 0) { s->err = 1; return -1; }
            mc_block(&d->dpb[di], c, pw, ph, xb, yb, bw, bh, m->mv[l][0], m->mv[l][1], p[l]);
        }
        uint8_t *dst = d->cur->pl[c] + yb * d->cur->stride[c] + xb; int stride = d->cur->stride[c];
        if (!wp) {                                                      /* 8.5.3.3.4.2 default weighted sample prediction */
            for (int y = 0; y < bh; y++) for (int x = 0; x < bw; x++) {
                int i = y * bw + x;
                dst[y * stride + x] = (uint8_t)(m->pf == 3 ? h_clip1((p[0][i] + p[1][i] + 64) >> 7) : h_clip1((p[m->pf == 2][i] + 32) >> 6));
            }
        } else {                                                        /* 8.5.3.3.4.3 explicit */
            int log2wd = (c ? sh->wp_log2_denom_c : sh->wp_log2_denom_l) + 6;
            int w0 = 0, w1 = 0, o0 = 0, o1 = 0;
            if (m->pf & 1) { w0 = sh->wp_w[0][m->ref[0]][c]; o0 = sh->wp_o[0][m->ref[0]][c]; }
            if (m->pf & 2) { w1 = sh->wp_w[1][m->ref[1]][c]; o1 = sh->wp_o[1][m->ref[1]][c]; }
            d->stats[HST_WP]++;
            for (int y = 0; y < bh; y++) for (int x = 0; x < bw; x++) {
                int i = y * bw + x, v;
                if (m->pf == 3) v = (p[0][i] * w0 + p[1][i] * w1 + ((o0 + o1 + 1) << log2wd)) >> (log2wd + 1);
                else if (m->pf == 1) v = ((p[0][i] * w0 + (1 << (log2wd - 1))) >> log2wd) + o0;
                else v = ((p[1][i] * w1 + (1 << (log2wd - 1))) >> log2wd) + o1;
                dst[y * stride + x] = (uint8_t)h_clip1(v);
            }
        }
    }
    return 0;
}

/* ------------------------------------------ 7.3.8.6 prediction_unit ------------------------------------------ */
static int parse_mvd(Sx *s, int16_t mvd[2]) {                          /* 7.3.8.9 */
    int g0[2], g1[2] = {0, 0};
    g0[0] = ae(s, ORCH_CTX_MVD_G0); g0[1] = ae(s, ORCH_CTX_MVD_G0);
    if (g0[0]) g1[0] = ae(s, ORCH_CTX_MVD_G1);
    if (g0[1]) g1[1] = ae(s, ORCH_CTX_MVD_G1);
    for (int k = 0; k < 2; k++) {
        int v = 0;
        if (g0[k]) {
            v = 1;
            if (g1[k]) {                                                /* abs_mvd_minus2: EG1 */
                int kk = 1, a = 0;
                while (ae_bypass(s)) { a += 1 << kk; kk++; if (kk > 17) { s->err = 1; return -1; } }
                a += ae_bypass_n(s, kk);
                v = a + 2;
            }
            if (ae_bypass(s)) v = -v;
        }
        if (v < -32768 || v > 32767) { s->err = 1; return -1; }
        mvd[k] = (int16_t)v;
    }
    return 0;
}
static int prediction_unit(Sx *s, int xcb, int ycb, int ncbs, int x0, int y0, int w, int h, int part_idx) {
    OrchDec *d = s->d; HSlice *sh = s->sh;
    Cand m; memset(&m, 0, sizeof m); m.ref[0] = m.ref[1] = -1;
    int merge = s->cu_skip ? 1 : ae(s, ORCH_CTX_MERGE_FLAG);
    s->last_merge = merge;
    if (merge) {
        int idx = 0;
        if (sh->max_merge_cand > 1) { idx = ae(s, ORCH_CTX_MERGE_IDX); if (idx) while (idx < sh->max_merge_cand - 1 && ae_bypass(s)) idx++; }
        m = merge_cand(s, xcb, ycb, ncbs, x0, y0, w, h, part_idx, idx);
        d->stats[HST_MERGE_PU]++;
    } else {
        int idc = 0;                                                    /* 0 = PRED_L0, 1 = PRED_L1, 2 = PRED_BI */
        if (sh->type == H_SLICE_B) {
            if (w + h != 12 && ae(s, ORCH_CTX_INTER_PRED_IDC + d->ct_depth[I4(d, x0, y0)])) idc = 2;
            else idc = ae(s, ORCH_CTX_INTER_PRED_IDC + 4);
        }
        int16_t mvd[2][2] = {{0, 0}, {0, 0}}; int mvp[2] = {0, 0};
        for (int l = 0; l < 2; l++) {
            if (idc == (l ? 0 : 1)) continue;
            int ri = 0;
            if (sh->n_ref[l] > 1) {                                     /* ref_idx_lX: TR, two context-coded bins */
                int cmax = sh->n_ref[l] - 1;
                while (ri < cmax && (ri < 2 ? ae(s, ORCH_CTX_REF_IDX + ri) : ae_bypass(s))) ri++;
            }
            m.ref[l] = (int8_t)ri; m.pf |= (uint8_t)(1 << l);
            if (l == 1 && sh->mvd_l1_zero && idc == 2) { mvd[1][0] = mvd[1][1] = 0; }
            else if (parse_mvd(s, mvd[l]) < 0) return -1;
            mvp[l] = ae(s, ORCH_CTX_MVP_FLAG);
        }
        for (int l = 0; l < 2; l++) if (m.pf >> l & 1) {
            int16_t pmv[2];
            amvp(s, xcb, ycb, ncbs, x0, y0, w, h, part_idx, l, m.ref[l], mvp[l], pmv);
            m.mv[l][0] = (int16_t)(pmv[0] + mvd[l][0]); m.mv[l][1] = (int16_t)(pmv[1] + mvd[l][1]);       /* modulo 2^16 (8-94 .. 8-97) */
        }
        d->stats[HST_AMVP_PU]++;
    }
    for (int l = 0; l < 2; l++) if ((m.pf >> l & 1) && (m.ref[l] < 0 || m.ref[l] >= sh->n_ref[l])) { s->err = 1; return -1; }
    if (m.pf == 3) d->stats[HST_BI_PU]++;
    for (int y = y0; y < y0 + h; y += 4) for (int x = x0; x < x0 + w; x += 4) {
        HMotion *o = &d->mot[I4(d, x, y)];
        memcpy(o->mv, m.mv, sizeof o->mv); o->ref_idx[0] = m.ref[0]; o->ref_idx[1] = m.ref[1]; o->pred_flag = m.pf;
        if (x == x0) d->edge[I4(d, x, y)] |= 4;
        if (y == y0) d->edge[I4(d, x, y)] |= 8;
    }
    if (d->digest_on) { dg(d, 0x5000 | (merge << 4) | m.pf); dg(d, x0); dg(d, y0); dg(d, w); dg(d, h); dg(d, m.ref[0]); dg(d, m.ref[1]); dg(d, m.mv[0][0]);
        dg(d, m.mv[0][1]); dg(d, m.mv[1][0]); dg(d, m.mv[1][1]); }
    return inter_pred(s, x0, y0, w, h, &m);
}

/* ------------------------------------------ 8.6.1 quantization parameters ------------------------------------------ */
static void derive_qp(Sx *s, int xcb, int ycb) {
    OrchDec *d = s->d;
    int lq = s->sps->log2_ctb - s->pps->diff_cu_qp_delta_depth;
    int xqg = xcb & ~((1 << lq) - 1), yqg = ycb & ~((1 << lq) - 1);
    int prev = s->first_qg ? s->sh->slice_qp : s->qp_y_prev;
    int a = prev, b = prev;
    int ctb_mask = ~((1 << s->sps->log2_ctb) - 1);
    if (avail_zs(s, xcb, ycb, xqg - 1, yqg) && ((xqg - 1) & ctb_mask) == (xqg & ctb_mask) && (yqg & ctb_mask) == (ycb & ctb_mask)) a = d->qp_y[I4(d, xqg - 1,
        yqg)];
    if (avail_zs(s, xcb, ycb, xqg, yqg - 1) && ((yqg - 1) & ctb_mask) == (yqg & ctb_mask) && (xqg & ctb_mask) == (xcb & ctb_mask)) b = d->qp_y[I4(d, xqg,
        yqg - 1)];
    int pred = (a + b + 1) >> 1;
    s->qp_y = ((pred + s->dqp + 52) % 52);
}
static int chroma_qp(const Sx *s, int c) {
    int off = c == 1 ? s->pps->cb_qp_offset + s->sh->cb_qp_offset : s->pps->cr_qp_offset + s->sh->cr_qp_offset;
    return orch_qpc_tab[h_clip3(0, 57, s->qp_y + off)];
}

/* ------------------------------------------ 7.3.8.8 transform_tree / 7.3.8.10 transform_unit ------------------------------------------ */
static int transform_unit(Sx *s, int x0, int y0, int xbase, int ybase, int log2, int depth, int blk, int cbf_y, int cbf_cb, int cbf_cr) {
    OrchDec *d = s->d;
    int n = 1 << log2;
    for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) {
        if (x == x0) d->edge[I4(d, x, y)] |= 1;
        if (y == y0) d->edge[I4(d, x, y)] |= 2;
        d->cbf[I4(d, x, y)] = (uint8_t)cbf_y;
    }
    d->stats[HST_TU4 + log2 - 2]++;
    if (cbf_y || cbf_cb || cbf_cr) {
        if (s->pps->cu_qp_delta && !s->is_dqp_coded) {
            int v = 0;                                                  /* cu_qp_delta_abs: prefix TU(5) + EG0 suffix */
            if (ae(s, ORCH_CTX_CU_QP_DELTA)) { v = 1; while (v < 5 && ae(s, ORCH_CTX_CU_QP_DELTA + 1)) v++; }
            if (v == 5) { int k = 0, a = 0; while (ae_bypass(s)) { a += 1 << k; k++; if (k > 16) { s->err = 1; return -1; } } a += ae_bypass_n(s, k); v += a; }
            if (v && ae_bypass(s)) v = -v;
            if (v < -26 || v > 25) { s->err = 1; return -1; }
            s->is_dqp_coded = 1; s->dqp = v;
            derive_qp(s, s->cu_x, s->cu_y);
            d->stats[HST_DQP]++;
        }
    }
    int tskip = 0;
    if (s->cu_intra) intra_pred(s, x0, y0, log2, 0, d->ipm[I4(d, x0, y0)]);
    if (cbf_y) {
        if (residual_coding(s, x0, y0, log2, 0, &tskip) < 0) return -1;
        if (s->cu_intra && n == 4) d->stats[HST_DST]++;
        residual_add(s, x0, y0, log2, 0, tskip, s->qp_y);
    }
    if (log2 > 2) {
        for (int c = 1; c < 3; c++) {
            if (s->cu_intra) intra_pred(s, x0 >> 1, y0 >> 1, log2 - 1, c, s->ipm_c);
            if (c == 1 ? cbf_cb : cbf_cr) { if (residual_coding(s, x0, y0, log2 - 1, c, &tskip) < 0) return -1;
                residual_add(s, x0 >> 1, y0 >> 1, log2 - 1, c, tskip, chroma_qp(s, c)); }
        }
    } else if (blk == 3) {
        for (int c = 1; c < 3; c++) {
            if (s->cu_intra) intra_pred(s, xbase >> 1, ybase >> 1, 2, c, s->ipm_c);
            if (c == 1 ? cbf_cb : cbf_cr) { if (residual_coding(s, xbase, ybase, 2, c, &tskip) < 0) return -1;
                residual_add(s, xbase >> 1, ybase >> 1, 2, c, tskip, chroma_qp(s, c)); }
        }
    }
    return 0;
}
static int transform_tree(Sx *s, int x0, int y0, int xbase, int ybase, int log2, int depth, int blk, int pcbf_cb, int pcbf_cr) {
    int split;
    if (log2 <= s->sps->log2_max_tb && log2 > s->sps->log2_min_tb && depth < s->max_tr_depth && !(s->intra_split && depth == 0)) split = ae(s,
        ORCH_CTX_SPLIT_TF + 5 - log2);
    else {
        int inter_split = s->sps->max_th_depth_inter == 0 && !s->cu_intra && s->part_mode != H_PART_2Nx2N && depth == 0;
        split = log2 > s->sps->log2_max_tb || (s->intra_split && depth == 0) || inter_split;
    }
    int cbf_cb = 0, cbf_cr = 0;
    if (log2 > 2) {
        if (pcbf_cb) cbf_cb = ae(s, ORCH_CTX_CBF_CBCR + depth);
        if (pcbf_cr) cbf_cr = ae(s, ORCH_CTX_CBF_CBCR + depth);
    } else { cbf_cb = pcbf_cb; cbf_cr = pcbf_cr; }                      /* 4x4 luma blocks: chroma flags of the parent (7.4.9.8) */
    if (split) {
        int h = 1 << (log2 - 1);
        for (int k = 0; k < 4; k++) if (transform_tree(s, x0 + (k & 1) * h, y0 + (k >> 1) * h, x0, y0, log2 - 1, depth + 1, k, cbf_cb, cbf_cr) < 0) return -1;
        return 0;
    }
    int cbf_y = 1;
    if (s->cu_intra || depth != 0 || cbf_cb || cbf_cr) cbf_y = ae(s, ORCH_CTX_CBF_LUMA + (depth == 0 ? 1 : 0));
    return transform_unit(s, x0, y0, xbase, ybase, log2, depth, blk, cbf_y, cbf_cb, cbf_cr);
}

/* ------------------------------------------ 7.3.8.5 coding_unit ------------------------------------------ */
static int coding_unit(Sx *s, int x0, int y0, int log2) {
    OrchDec *d = s->d; HSlice *sh = s->sh; const HSps *sps = s->sps;
    const int n = 1 << log2;
    s->cu_x = x0; s->cu_y = y0; s->cu_log2 = log2; s->cu_intra = 0; s->cu_skip = 0; s->part_mode = H_PART_2Nx2N; s->tq_bypass = 0; s->intra_split = 0;
    d->stats[HST_CU]++;
    if (s->pps->tq_bypass) { s->tq_bypass = ae(s, ORCH_CTX_CU_TQ_BYPASS); if (s->tq_bypass) d->stats[HST_BYPASS]++; }
    if (sh->type != H_SLICE_I) {
        int inc = 0;
        if (avail_zs(s, x0, y0, x0 - 1, y0) && d->skip_flag[I4(d, x0 - 1, y0)]) inc++;
        if (avail_zs(s, x0, y0, x0, y0 - 1) && d->skip_flag[I4(d, x0, y0 - 1)]) inc++;
        s->cu_skip = ae(s, ORCH_CTX_CU_SKIP + inc);
    }
    int pcm = 0, rqt_root_cbf = 1;
    if (!s->cu_skip) {
        if (sh->type != H_SLICE_I) s->cu_intra = ae(s, ORCH_CTX_PRED_MODE); else s->cu_intra = 1;
        if (!s->cu_intra || log2 == sps->log2_min_cb) {                /* part_mode (9.3.3.5) */
            if (s->cu_intra) s->part_mode = ae(s, ORCH_CTX_PART_MODE) ? H_PART_2Nx2N : H_PART_NxN;
            else if (ae(s, ORCH_CTX_PART_MODE)) s->part_mode = H_PART_2Nx2N;
            else if (log2 == sps->log2_min_cb) {
                if (ae(s, ORCH_CTX_PART_MODE + 1)) s->part_mode = H_PART_2NxN;
                else if (log2 == 3) s->part_mode = H_PART_Nx2N;
                else s->part_mode = ae(s, ORCH_CTX_PART_MODE + 2) ? H_PART_Nx2N : H_PART_NxN;
            } else if (!sps->amp) s->part_mode = ae(s, ORCH_CTX_PART_MODE + 1) ? H_PART_2NxN : H_PART_Nx2N;
            else {
                int hor = ae(s, ORCH_CTX_PART_MODE + 1);
                if (ae(s, ORCH_CTX_PART_MODE + 3)) s->part_mode = hor ? H_PART_2NxN : H_PART_Nx2N;
                else { int b = ae_bypass(s); s->part_mode = hor ? (b ? H_PART_2NxnD : H_PART_2NxnU) : (b ? H_PART_nRx2N : H_PART_nLx2N); d->stats[HST_AMP]++; }
            }
        }
    }
    /* CU-level maps that the syntax below (and later CUs) read */
    for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) {
        int i = I4(d, x, y);
        d->pred_mode[i] = (uint8_t)(s->cu_intra ? 2 : 1); d->skip_flag[i] = (uint8_t)s->cu_skip; d->nofilter[i] = (uint8_t)s->tq_bypass;
        d->slice_of4[i] = (int16_t)s->slice_idx;
        d->edge[i] = 0; d->cbf[i] = 0; d->ipm[i] = 1;
        memset(&d->mot[i], 0, sizeof d->mot[i]); d->mot[i].ref_idx[0] = d->mot[i].ref_idx[1] = -1;
        if (x == x0) d->edge[i] |= 1 | 4;
        if (y == y0) d->edge[i] |= 2 | 8;
    }
    if (d->digest_on) { dg(d, 0x4000 | (s->cu_skip << 8) | (s->cu_intra << 7) | (s->tq_bypass << 6) | (s->part_mode << 3) | log2); dg(d, x0); dg(d, y0); }
    if (s->cu_skip) { d->stats[HST_SKIP_CU]++; if (prediction_unit(s, x0, y0, n, x0, y0, n, n, 0) < 0) return -1; }
    else if (s->cu_intra) {
        d->stats[HST_INTRA_CU]++;
        if (s->part_mode == H_PART_2Nx2N && sps->pcm && log2 >= sps->log2_min_pcm && log2 <= sps->log2_max_pcm) pcm = ae_terminate(s);
        if (pcm) {
            d->stats[HST_PCM]++;
            cabac_align(s);
            for (int c = 0; c < 3; c++) {
                int sc = c ? 1 : 0, nn = n >> sc, bits = c ? sps->pcm_bits_c : sps->pcm_bits_y;
                uint8_t *dst = d->cur->pl[c] + (y0 >> sc) * d->cur->stride[c] + (x0 >> sc);
                for (int y = 0; y < nn; y++) for (int x = 0; x < nn; x++) { int v = (int)bits_u(&s->b, bits);
                    dst[y * d->cur->stride[c] + x] = (uint8_t)(v << (8 - bits)); if (d->digest_on) dg(d, v); }
            }
            cabac_init_engine(s);
            if (sps->pcm_loop_filter_disabled) for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) d->nofilter[I4(d, x, y)] = 1;
            rqt_root_cbf = 0;
        } else {
            int np = s->part_mode == H_PART_NxN ? 2 : 1, pb = n / np;
            if (np == 2) { s->intra_split = 1; d->stats[HST_NXN]++; }
            int prev_flag[4], modes[4];
            for (int k = 0; k < np * np; k++) prev_flag[k] = ae(s, ORCH_CTX_PREV_INTRA);
            for (int k = 0; k < np * np; k++) {
                int xp = x0 + (k & 1) * pb, yp = y0 + (k >> 1) * pb;
                int idx;
                if (prev_flag[k]) { idx = 0; while (idx < 2 && ae_bypass(s)) idx++; } else idx = ae_bypass_n(s, 5);
                /* 8.4.2 derivation of the luma intra prediction mode */
                int ca = 1, cb = 1;
                if (avail_zs(s, xp, yp, xp - 1, yp) && d->pred_mode[I4(d, xp - 1, yp)] == 2) ca = d->ipm[I4(d, xp - 1, yp)];
                if (avail_zs(s, xp, yp, xp, yp - 1) && d->pred_mode[I4(d, xp, yp - 1)] == 2 &&
                    yp - 1 >= ((yp >> sps->log2_ctb) << sps->log2_ctb)) cb = d->ipm[I4(d, xp, yp - 1)];
                int cand[3];
                if (ca == cb) { if (ca < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; } else { cand[0] = ca; cand[1] = 2 + ((ca + 29) % 32);
                    cand[2] = 2 + ((ca - 2 + 1) % 32); } }
                else { cand[0] = ca; cand[1] = cb; cand[2] = (ca != 0 && cb != 0) ? 0 : ((ca != 1 && cb != 1) ? 1 : 26); }
                int mode;
                if (prev_flag[k]) mode = cand[idx];
                else {
                    int t;
                    if (cand[0] > cand[1]) { t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
                    if (cand[0] > cand[2]) { t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
                    if (cand[1] > cand[2]) { t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
                    mode = idx;
                    for (int i = 0; i < 3; i++) if (mode >= cand[i]) mode++;
                }
                modes[k] = mode;
                for (int y = yp; y < yp + pb; y += 4) for (int x = xp; x < xp + pb; x += 4) d->ipm[I4(d, x, y)] = (uint8_t)mode;
            }
            int cm = 4;                                                 /* intra_chroma_pred_mode */
            if (ae(s, ORCH_CTX_INTRA_CHROMA)) cm = ae_bypass_n(s, 2);
            static const uint8_t ctab[4] = {0, 26, 10, 1};
            s->ipm_c = cm == 4 ? modes[0] : (ctab[cm] == modes[0] ? 34 : ctab[cm]);
            if (d->digest_on) { for (int k = 0; k < np * np; k++) dg(d, modes[k]); dg(d, s->ipm_c); }
        }
    } else {
        int w[4], h[4], xs[4], ys[4], np = 2;
        xs[0] = x0; ys[0] = y0;
        switch (s->part_mode) {
        case H_PART_2Nx2N: np = 1; w[0] = n; h[0] = n; break;
        case H_PART_2NxN: w[0] = w[1] = n; h[0] = h[1] = n / 2; xs[1] = x0; ys[1] = y0 + n / 2; break;
        case H_PART_Nx2N: w[0] = w[1] = n / 2; h[0] = h[1] = n; xs[1] = x0 + n / 2; ys[1] = y0; break;
        case H_PART_2NxnU: w[0] = w[1] = n; h[0] = n / 4; h[1] = n * 3 / 4; xs[1] = x0; ys[1] = y0 + n / 4; break;
        case H_PART_2NxnD: w[0] = w[1] = n; h[0] = n * 3 / 4; h[1] = n / 4; xs[1] = x0; ys[1] = y0 + n * 3 / 4; break;
        case H_PART_nLx2N: h[0] = h[1] = n; w[0] = n / 4; w[1] = n * 3 / 4; xs[1] = x0 + n / 4; ys[1] = y0; break;
        case H_PART_nRx2N: h[0] = h[1] = n; w[0] = n * 3 / 4; w[1] = n / 4; xs[1] = x0 + n * 3 / 4; ys[1] = y0; break;
        default: np = 4; for (int k = 0; k < 4; k++) { w[k] = h[k] = n / 2; xs[k] = x0 + (k & 1) * n / 2; ys[k] = y0 + (k >> 1) * n / 2; } d->stats[HST_NXN]++;
        break;
        }
        for (int k = 0; k < np; k++) if (prediction_unit(s, x0, y0, n, xs[k], ys[k], w[k], h[k], k) < 0) return -1;
    }
    if (s->err) return -1;
    if (!pcm && !s->cu_skip) {
        if (!s->cu_intra && !(s->part_mode == H_PART_2Nx2N && s->last_merge)) rqt_root_cbf = ae(s, ORCH_CTX_RQT_ROOT_CBF);
        if (rqt_root_cbf) {
            s->max_tr_depth = s->cu_intra ? sps->max_th_depth_intra + s->intra_split : sps->max_th_depth_inter;
            if (transform_tree(s, x0, y0, x0, y0, log2, 0, 0, 1, 1) < 0) return -1;
        }
    }
    if (s->cu_intra && !pcm && !rqt_root_cbf) { s->err = 1; return -1; }
    /* 8.6.1: QpY of the coding unit (after a cu_qp_delta inside it took effect) */
    for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) d->qp_y[I4(d, x, y)] = (int8_t)s->qp_y;
    d->last_cu_qp = s->qp_y; s->qg_started = 1;
    TRACE("CU %d %d %d qp %d\n", x0, y0, log2, s->qp_y);
    if (d->digest_on) dg(d, 0x4800 | s->qp_y);
    return s->err ? -1 : 0;
}

/* ------------------------------------------ 7.3.8.4 coding_quadtree ------------------------------------------ */
static int coding_quadtree(Sx *s, int x0, int y0, int log2, int depth) {
    OrchDec *d = s->d; const HSps *sps = s->sps;
    int n = 1 << log2, split;
    if (x0 + n <= d->w && y0 + n <= d->h && log2 > sps->log2_min_cb) {
        int inc = 0;
        if (avail_zs(s, x0, y0, x0 - 1, y0) && d->ct_depth[I4(d, x0 - 1, y0)] > depth) inc++;
        if (avail_zs(s, x0, y0, x0, y0 - 1) && d->ct_depth[I4(d, x0, y0 - 1)] > depth) inc++;
        split = ae(s, ORCH_CTX_SPLIT_CU + inc);
    } else split = log2 > sps->log2_min_cb;
    if (s->pps->cu_qp_delta && log2 >= sps->log2_ctb - s->pps->diff_cu_qp_delta_depth) {       /* start of a quantization group */
        s->is_dqp_coded = 0; s->dqp = 0;
        if (s->qg_started) { s->qp_y_prev = d->last_cu_qp; s->first_qg = 0; }     /* qg_started: a coding unit was decoded since the last reset */
    }
    if (split) {
        int h = n >> 1;
        for (int k = 0; k < 4; k++) {
            int x = x0 + (k & 1) * h, y = y0 + (k >> 1) * h;
            if (x < d->w && y < d->h && coding_quadtree(s, x, y, log2 - 1, depth + 1) < 0) return -1;
        }
        return 0;
    }
    for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) d->ct_depth[I4(d, x, y)] = (uint8_t)depth;
    derive_qp(s, x0, y0);
    return coding_unit(s, x0, y0, log2);
}

/* ------------------------------------------ 7.3.8.3 sao ------------------------------------------ */
static void parse_sao(Sx *s, int rx, int ry) {
    OrchDec *d = s->d; HSlice *sh = s->sh;
    HSao *o = &d->sao[s->ctb_addr_rs];
    memset(o, 0, sizeof *o);
    if (!sh->sao_luma && !sh->sao_chroma) return;
    int merge_left = 0, merge_up = 0;
    if (rx > 0) {
        int left_in_slice = d->ctb_slice_addr[s->ctb_addr_rs - 1] == sh->slice_addr,
            left_in_tile = d->tile_id[d->ctb_rs2ts[s->ctb_addr_rs - 1]] == d->tile_id[s->ctb_addr_ts];
        if (left_in_slice && left_in_tile) merge_left = ae(s, ORCH_CTX_SAO_MERGE);
    }
    if (ry > 0 && !merge_left) {
        int up = s->ctb_addr_rs - d->ctb_w;
        if (d->ctb_slice_addr[up] == sh->slice_addr && d->tile_id[d->ctb_rs2ts[up]] == d->tile_id[s->ctb_addr_ts]) merge_up = ae(s, ORCH_CTX_SAO_MERGE);
    }
    if (merge_left) { *o = d->sao[s->ctb_addr_rs - 1]; }
    else if (merge_up) { *o = d->sao[s->ctb_addr_rs - d->ctb_w]; }
    else for (int c = 0; c < 3; c++) {
        if (!(c == 0 ? sh->sao_luma : sh->sao_chroma)) continue;
        if (c == 2) { o->type[2] = o->type[1]; }
        else { int t = 0; if (ae(s, ORCH_CTX_SAO_TYPE)) t = ae_bypass(s) ? 2 : 1; o->type[c] = (uint8_t)t; }
        if (!o->type[c]) continue;
        int a[4];
        for (int i = 0; i < 4; i++) { a[i] = 0; while (a[i] < 7 && ae_bypass(s)) a[i]++; }
        if (o->type[c] == 1) {
            for (int i = 0; i < 4; i++) if (a[i] && ae_bypass(s)) a[i] = -a[i];
            o->band_pos[c] = (uint8_t)ae_bypass_n(s, 5);
            d->stats[HST_SAO_BAND]++;
        } else {
            a[2] = -a[2]; a[3] = -a[3];
            if (c == 0) o->eo_class[0] = (uint8_t)ae_bypass_n(s, 2);
            else if (c == 1) o->eo_class[1] = (uint8_t)ae_bypass_n(s, 2);
            else o->eo_class[2] = o->eo_class[1];
            d->stats[HST_SAO_EDGE]++;
        }
        for (int i = 0; i < 4; i++) o->off[c][i] = (int8_t)a[i];
    }
    /* components whose slice flag is off are not filtered even when the parameters were merged from a neighbour */
    if (!sh->sao_luma) o->type[0] = 0;
    if (!sh->sao_chroma) o->type[1] = o->type[2] = 0;
    if (d->digest_on) for (int c = 0; c < 3; c++) { dg(d, 0x6000 | (c << 8) | (o->type[c] << 6) | (o->type[c] == 1 ? o->band_pos[c] : o->eo_class[c]));
        if (o->type[c]) for (int i = 0; i < 4; i++) dg(d, o->off[c][i]); }
}

/* ------------------------------------------ 7.3.8.1 slice_segment_data ------------------------------------------ */
int orch_decode_slice_data(OrchDec *d, HSlice *sh, int slice_idx, const uint8_t *rbsp, size_t len) {
    static __thread Sx sx;
    Sx *s = &sx;
    memset(s, 0, sizeof *s);
    s->d = d; s->sh = sh; s->sps = d->asps; s->pps = d->apps; s->slice_idx = slice_idx;
    if (sh->data_offset >= len) H_FAIL(d, "slice segment without data");
    bits_init(&s->b, rbsp + sh->data_offset, len - sh->data_offset);
    const int n_ctb = d->ctb_w * d->ctb_h;
    s->ctb_addr_rs = sh->segment_addr; s->ctb_addr_ts = d->ctb_rs2ts[s->ctb_addr_rs];
    if (sh->dependent) {
        if (!d->dep_valid) H_FAIL(d, "dependent slice segment without stored context variables");
        memcpy(s->st, d->dep_st, sizeof s->st); memcpy(s->mps, d->dep_mps, sizeof s->mps);
        s->qp_y_prev = d->last_cu_qp; s->first_qg = 0; s->qg_started = 0;
        d->stats[HST_DEP_SLICE]++;
    } else { cabac_init_ctx(s); s->first_qg = 1; s->qg_started = 0; s->qp_y_prev = sh->slice_qp; }
    s->qp_y = sh->slice_qp;
    cabac_init_engine(s);
    int first_ctu = 1;
    for (;;) {
        int rx = s->ctb_addr_rs % d->ctb_w, ry = s->ctb_addr_rs / d->ctb_w;
        int tile = d->tile_id[s->ctb_addr_ts];
        int first_in_tile = s->ctb_addr_ts == 0 || d->tile_id[s->ctb_addr_ts - 1] != tile;
        int row_start = s->pps->wpp && (rx == 0 || d->tile_id[d->ctb_rs2ts[s->ctb_addr_rs - 1]] != tile);
        d->ctb_slice_addr[s->ctb_addr_rs] = sh->slice_addr; d->ctb_slice_idx[s->ctb_addr_rs] = (int16_t)slice_idx;
        /* 9.3.1: "If the CTU is the first CTU in a tile, the initialization process is invoked" comes BEFORE the dependent-slice-segment case:
         * a dependent segment that opens a tile does not inherit the previous segment's context variables */
        if (first_in_tile && first_ctu && sh->dependent) d->stats[HST_DEP_OPENS_TILE]++;
        if (first_in_tile) { if (!first_ctu || sh->dependent) cabac_init_ctx(s); s->first_qg = 1; s->qg_started = 0; s->qp_y_prev = sh->slice_qp; }
        else if (row_start) {                                           /* 9.3.1: synchronisation with the CTB above right */
            int x0 = rx << s->sps->log2_ctb, y0 = ry << s->sps->log2_ctb;
            int avail_t = avail_zs(s, x0, y0, x0 + d->ctb_size, y0 - d->ctb_size);
            if (avail_t && d->wpp_valid_pic) { memcpy(s->st, d->wpp_st, sizeof s->st); memcpy(s->mps, d->wpp_mps, sizeof s->mps); }
            else if (!first_ctu) cabac_init_ctx(s);
            s->first_qg = 1; s->qg_started = 0; s->qp_y_prev = sh->slice_qp;
            d->stats[HST_WPP_ROWS]++;
        }
        first_ctu = 0;
        parse_sao(s, rx, ry);
        if (coding_quadtree(s, rx << s->sps->log2_ctb, ry << s->sps->log2_ctb, s->sps->log2_ctb, 0) < 0 || s->err || s->b.err)
            H_FAIL(d, "corrupt slice data at CTB %d (POC %d)", s->ctb_addr_rs, d->cur->poc);
        if (s->pps->wpp) {                                              /* 9.3.2.2 storage after the 2nd CTB of a row (of a tile) */
            int second = rx == 1 || (s->ctb_addr_rs > 1 && rx > 1 && d->tile_id[d->ctb_rs2ts[s->ctb_addr_rs - 2]] != tile);
            if (second) { memcpy(d->wpp_st, s->st, sizeof s->st); memcpy(d->wpp_mps, s->mps, sizeof s->mps); d->wpp_valid_pic = 1; }
        }
        int end = ae_terminate(s);
        s->ctb_addr_ts++;
        if (end) break;
        if (s->ctb_addr_ts >= n_ctb) H_FAIL(d, "slice data runs past the last CTB");
        s->ctb_addr_rs = d->ctb_ts2rs[s->ctb_addr_ts];
        int new_tile = s->pps->tiles && d->tile_id[s->ctb_addr_ts] != d->tile_id[s->ctb_addr_ts - 1];
        int new_row = s->pps->wpp && (s->ctb_addr_rs % d->ctb_w == 0 || d->tile_id[s->ctb_addr_ts] != d->tile_id[d->ctb_rs2ts[s->ctb_addr_rs - 1]]);
        if (new_tile || new_row) {
            if (!ae_terminate(s)) H_FAIL(d, "end_of_subset_one_bit missing");
            cabac_align(s);
            cabac_init_engine(s);
        }
    }
    if (s->pps->dependent_slices) { memcpy(d->dep_st, s->st, sizeof s->st); memcpy(d->dep_mps, s->mps, sizeof s->mps); d->dep_valid = 1; }
    if (s->err || s->b.err) H_FAIL(d, "corrupt slice data (POC %d)", d->cur->poc);
    return 0;
}
