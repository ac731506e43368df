/*
 * tools/hevcgen.c -- seeded synthetic HEVC (ITU-T H.265 Main profile, 8-bit 4:2:0) Annex-B stream generator.
 *
 * The reference repository ships no bitstreams and this image has no HEVC encoder, so every HEVC
 * test and bench input is produced here: a small closed-loop encoder (CABAC, I/P/B with reference picture sets, coding
 * quadtree down to 8x8, 2Nx2N / 2NxN / Nx2N / NxN / AMP partitions, merge + AMVP + temporal candidates, 35 intra modes,
 * residual quadtree with 4x4 .. 32x32 transforms, DST, transform skip, sign data hiding, cu_qp_delta, PCM, transquant
 * bypass, scaling lists, weighted prediction, deblocking, SAO, slices / dependent slice segments / tiles / wavefronts)
 * with its OWN reconstruction loop.  The reconstruction written with --recon must equal what a conforming decoder
 * outputs, which gives the CPU oracle (oracle/orc_hevc_*.c) and the HIP decoder a second, independently written code
 * path to agree with (tests/test_hevc_oracle.py).  Decisions are a mix of SAD heuristics and seeded randomness: the aim
 * is syntax coverage and realistic statistics, not compression.
 *
 * Not derived from the reference (which contains no codec arithmetic).  Content is procedural and integer-only so that
 * streams are bit-identical on every machine: seed = 0x4A4D0000 + config_id*256 + stream_id.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hevcgen_tables.h"

#define CLIP3(lo, hi, v) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))
#define CLIP1(v) CLIP3(0, 255, v)
#define ABS(v) ((v) < 0 ? -(v) : (v))
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define SIGN(v) ((v) < 0 ? -1 : ((v) > 0 ? 1 : 0))

typedef struct {
    int width, height;          /* display size (conformance window, origin 0,0)                          */
    int frames, qp, seed;
    int intra_period;           /* IDR every n frames (display order)                                       */
    int gop;                    /* 0: I P P P ..; 1..3: n non-reference B pictures between anchors; 8: hierarchical random-access GOP of 8 */
    int num_ref;                /* active references per list (1..4) for P / anchor pictures                */
    int ctb_log2;               /* 4..6                                                                     */
    int min_cb_log2;            /* 3..ctb_log2                                                              */
    int max_tb_log2, min_tb_log2, depth_inter, depth_intra;
    int mode;                   /* 0 = SAD decisions on procedural content, 1 = fuzz (random decisions)     */
    int amp, sao, deblock;      /* deblock: 1 on, 0 off (PPS), 2 on with slice-level overrides / offsets    */
    int tskip, sdh, dqp, pcm, bypass, cip, strong_intra, tmvp, wp, rplm, lt_ref;
    int scaling;                /* 0 off, 1 default lists, 2 lists in the SPS, 3 lists in the PPS           */
    int wpp, tile_cols, tile_rows;
    int slice_ctus;             /* > 0: a new slice segment every n CTUs                                    */
    int dep_slices;             /* 1: every second segment is a dependent slice segment                     */
    int merge_cand;             /* MaxNumMergeCand 1..5                                                     */
    int cabac_init;             /* cabac_init_flag in P/B slices                                            */
    int par_mrg;                /* Log2ParMrgLevel 2..6                                                     */
    int rps_sps;                /* 1: reference picture sets live in the SPS (inter RPS prediction where expressible) */
    int cb_qp_off, cr_qp_off;
    int search;                 /* integer search range (mode 0)                                            */
    int open_gop; /* 1: intra periods after the first start with a CRA picture whose leading B pictures are RASL (gop 1..3, period a multiple of gop + 1);
    parameter sets are repeated there */
    int vui_fps;                /* > 0: the SPS carries vui_parameters() with vui_timing_info (num_units_in_tick 1, time_scale vui_fps); no effect on decoding */
} HevcGenParams;

/* ------------------------------ RNG ------------------------------ */
typedef struct { uint64_t s; } Rng;
static uint32_t rnd(Rng *r) { r->s = r->s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(r->s >> 33); }
static int rnd_n(Rng *r, int n) { return (int)(rnd(r) % (uint32_t)n); }

/* ------------------------------ bit writer ------------------------------ */
typedef struct { uint8_t *buf; size_t cap, len; uint32_t cur; int nbits; } BitW;
static void bw_reserve(BitW *w, size_t extra) { if (w->len + extra > w->cap) { w->cap = (w->len + extra) * 2 + 1024;
    w->buf = (uint8_t *)realloc(w->buf, w->cap); } }
static void bw_put(BitW *w, int n, uint32_t v) {
    for (int i = n - 1; i >= 0; i--) {
        w->cur = (w->cur << 1) | ((v >> i) & 1);
        if (++w->nbits == 8) { bw_reserve(w, 1); w->buf[w->len++] = (uint8_t)w->cur; w->cur = 0; w->nbits = 0; }
    }
}
static void bw_ue(BitW *w, uint32_t v) { uint32_t x = v + 1; int n = 0; while ((x >> n) > 1) n++; bw_put(w, n, 0); bw_put(w, n + 1, x); }
static void bw_se(BitW *w, int v) { bw_ue(w, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
static void bw_trailing(BitW *w) { bw_put(w, 1, 1); while (w->nbits) bw_put(w, 1, 0); }
static void bw_bytes(BitW *w, const uint8_t *p, size_t n) { bw_reserve(w, n); memcpy(w->buf + w->len, p, n); w->len += n; }
static int ceil_log2(int v) { int n = 0; while ((1 << n) < v) n++; return n; }

/* NAL unit: start code + 2-byte header + payload with emulation prevention.  `marks` (optional): byte positions inside the
 * payload whose positions in the escaped output are wanted back (entry points, 7.4.7.1: offsets count the prevention bytes) */
static void write_nal(BitW *out, int type, int tid, const uint8_t *p, size_t n, size_t *marks, int n_marks) {
    static const uint8_t sc[4] = {0, 0, 0, 1};
    bw_bytes(out, sc, 4);
    uint8_t hdr[2] = { (uint8_t)(type << 1), (uint8_t)(tid + 1) };
    bw_bytes(out, hdr, 2);
    int zeros = 0, mk = 0; size_t base = out->len;
    for (size_t i = 0; i < n; i++) {
        while (mk < n_marks && marks[mk] == i) marks[mk++] = out->len - base;
        if (zeros >= 2 && p[i] <= 3) { bw_reserve(out, 1); out->buf[out->len++] = 3; zeros = 0; }
        bw_reserve(out, 1); out->buf[out->len++] = p[i];
        zeros = p[i] == 0 ? zeros + 1 : 0;
    }
    while (mk < n_marks) marks[mk++] = out->len - base;
}

/* ------------------------------ CABAC encoder (9.3.4.? mirrored: 9.3.5 in the encoder annex wording) ------------------------------ */
typedef struct {
    uint32_t low, range; int outstanding, first;
    BitW *w;
    uint8_t st[HG_N_CTX], mps[HG_N_CTX];
} Cab;
static void cab_init_ctx(Cab *c, int init_type, int qp) {
    qp = CLIP3(0, 51, qp);
    for (int i = 0; i < HG_N_CTX; i++) {
        int v = hg_ctx_init[init_type][i], m = (v >> 4) * 5 - 45, n = ((v & 15) << 3) - 16;
        int pre = CLIP3(1, 126, ((m * qp) >> 4) + n);
        c->mps[i] = pre > 63; c->st[i] = (uint8_t)(pre > 63 ? pre - 64 : 63 - pre);
    }
}
static void cab_start(Cab *c, BitW *w) { c->low = 0; c->range = 510; c->outstanding = 0; c->first = 1; c->w = w; }
static void cab_putbit(Cab *c, int b) {
    if (c->first) c->first = 0; else bw_put(c->w, 1, (uint32_t)b);
    while (c->outstanding > 0) { bw_put(c->w, 1, (uint32_t)!b); c->outstanding--; }
}
static void cab_renorm(Cab *c) {
    while (c->range < 256) {
        if (c->low < 256) cab_putbit(c, 0);
        else if (c->low >= 512) { c->low -= 512; cab_putbit(c, 1); }
        else { c->low -= 256; c->outstanding++; }
        c->range <<= 1; c->low <<= 1;
    }
}
static FILE *g_trace; static int g_trace_init;
#define TRACE(...) do { if (!g_trace_init) { g_trace_init = 1; if (getenv("HG_TRACE")) g_trace = fopen(getenv("HG_TRACE"), "w"); } \
        if (g_trace) fprintf(g_trace, __VA_ARGS__); } while (0)
static void cab_enc(Cab *c, int ctx, int bin) {
    TRACE("c%d %d\n", ctx, bin);
    uint32_t lps = hg_range_lps[c->st[ctx]][(c->range >> 6) & 3];
    c->range -= lps;
    if (bin != c->mps[ctx]) {
        c->low += c->range; c->range = lps;
        if (c->st[ctx] == 0) c->mps[ctx] ^= 1;
        c->st[ctx] = hg_trans_lps[c->st[ctx]];
    } else if (c->st[ctx] < 62) c->st[ctx]++;
    cab_renorm(c);
}
static void cab_byp(Cab *c, int bin) {
    TRACE("b %d\n", bin);
    c->low <<= 1;
    if (bin) c->low += c->range;
    if (c->low >= 1024) { cab_putbit(c, 1); c->low -= 1024; }
    else if (c->low < 512) cab_putbit(c, 0);
    else { c->low -= 512; c->outstanding++; }
}
static void cab_byp_n(Cab *c, int n, uint32_t v) { for (int i = n - 1; i >= 0; i--) cab_byp(c, (int)((v >> i) & 1)); }
static void cab_term(Cab *c, int bin) {
    TRACE("t %d\n", bin);
    c->range -= 2;
    if (bin) {                                      /* flush: the last bit written is the stop / alignment bit */
        c->low += c->range; c->range = 2; cab_renorm(c);
        cab_putbit(c, (int)((c->low >> 9) & 1));
        bw_put(c->w, 2, ((c->low >> 7) & 3) | 1);
    } else cab_renorm(c);
}
static void cab_egk(Cab *c, int k, uint32_t v) { while (v >= (1u << k)) { cab_byp(c, 1); v -= 1u << k; k++; } cab_byp(c, 0); cab_byp_n(c, k, v); }

/* ------------------------------ pictures ------------------------------ */
typedef struct { int16_t mv[2][2]; int8_t ref[2]; uint8_t pf; } Mot;
typedef struct {
    uint8_t *pl[3]; int stride[3];
    int poc, is_ref /* 0, 1 short, 2 long */, used, type, tid;
    Mot *col; int *col_poc; uint8_t *col_lt, *col_intra;          /* 16x16 compressed motion of the picture (temporal candidates) */
} Pic;

typedef struct { int n_neg, n_pos, dneg[16], uneg[16], dpos[16], upos[16]; } RpsSet;
typedef struct { int type[3], band[3], eo[3], off[3][4]; } Sao;
typedef struct {                                                    /* one slice (independent segment): what later derivations need */
    int addr, type, qp, deblock_off, beta, tc, lf_across, n_ref[2];
    int ref_poc[2][16], ref_lt[2][16]; Pic *ref[2][16];
    int wp_denom[2]; int wp_w[2][16][3], wp_o[2][16][3]; int wp_on;
    int tmvp, col_l0, col_idx, max_merge, mvd_l1_zero, cabac_init, sao_l, sao_c;
    int rplm_flag[2], list_entry[2][16];
} Slc;

typedef struct Enc {
    HevcGenParams p; Rng rng;
    int W, H, ctb, ctb_w, ctb_h, w4, h4, poc_bits;
    uint8_t *tex; int tex_w, tex_h;                                 /* procedural texture the frames are cut from */
    Pic src, dpb[10], *cur;
    /* 4x4-granular maps of the current picture */
    uint8_t *pm /* 0 none 1 inter 2 intra */, *skip, *depth, *ipm, *nofilt, *edges, *cbf; int8_t *qpmap; Mot *mot; int16_t *slice_of;
    int *ctb_slice;                                                 /* SliceAddrRs per CTB, -1 before it is coded */
    int *rs2ts, *ts2rs, *tile_of;                                   /* 6.5.1 */
    Sao *sao;
    Slc slices[512]; int n_slices; Slc *sl;
    Cab cab;
    /* per-CU state */
    int qp_cur, qp_prev, last_cu_qp, dqp_coded, dqp_val, first_qg, qg_open, qg_x, qg_y;
    int seg_first_ctb;                                              /* raster address of the first CTB of the current slice (SliceAddrRs) */
    int cur_ts;
    uint8_t sf[4][6][1024]; int sf_on;                              /* scaling factors m[x][y] by size id / matrix id */
    uint8_t *recon_buf; int recon_frames; FILE *recon;
    int decode_count;
    uint8_t *dbk[3];
    int lf_across_tiles;
    int tile_explicit, tile_cb[21], tile_rb[23]; /* tile boundaries in CTBs; explicit: sent as column widths / row heights (uniform_spacing_flag = 0) */
    RpsSet sps_sets[64]; int n_sps_sets;                               /* short-term reference picture sets carried by the SPS (rps_sps) */
    uint8_t sl4[6][16], sl8[6][64], sl16[6][64], sl32[6][64], dc16[6], dc32[6];   /* coded scaling lists (diagonal order) */
} Enc;
#define I4(e, x, y) (((y) >> 2) * (e)->w4 + ((x) >> 2))

static void pic_alloc(Enc *e, Pic *p) {
    for (int c = 0; c < 3; c++) { p->stride[c] = e->W >> (c ? 1 : 0); p->pl[c] = (uint8_t *)calloc((size_t)p->stride[c], (size_t)(e->H >> (c ? 1 : 0))); }
    size_t n = (size_t)((e->W + 15) >> 4) * (size_t)((e->H + 15) >> 4);
    p->col = (Mot *)calloc(n, sizeof(Mot)); p->col_poc = (int *)calloc(n * 2, sizeof(int)); p->col_lt = (uint8_t *)calloc(n, 1);
    p->col_intra = (uint8_t *)calloc(n, 1);
}

/* ------------------------------ procedural content ------------------------------ */
static void make_texture(Enc *e) {
    e->tex_w = e->W + 256; e->tex_h = e->H + 256;
    e->tex = (uint8_t *)malloc((size_t)e->tex_w * e->tex_h);
    Rng r = { (uint64_t)e->p.seed * 77 + 5 };
    int gw = e->tex_w / 32 + 2, gh = e->tex_h / 32 + 2;
    int *grid = (int *)malloc(sizeof(int) * (size_t)gw * gh);
    for (int i = 0; i < gw * gh; i++) grid[i] = 40 + rnd_n(&r, 176);
    for (int y = 0; y < e->tex_h; y++) for (int x = 0; x < e->tex_w; x++) {
        int gx = x >> 5, gy = y >> 5, fx = x & 31, fy = y & 31;
        int a = grid[gy * gw + gx], b = grid[gy * gw + gx + 1], c = grid[(gy + 1) * gw + gx], d = grid[(gy + 1) * gw + gx + 1];
        int v = ((a * (32 - fx) + b * fx) * (32 - fy) + (c * (32 - fx) + d * fx) * fy) >> 10;
        v += (int)(rnd(&r) % 9) - 4;                                   /* band-limited noise */
        if (((x >> 4) + (y >> 4)) % 7 == 0) v += ((x ^ y) & 8) ? 12 : -12;   /* some hard edges */
        e->tex[y * e->tex_w + x] = (uint8_t)CLIP1(v);
    }
    free(grid);
}
static void make_source(Enc *e, int t) {                                /* frame t (display order): global pan + two moving blocks */
    int ox = 64 + ((t * 3) % 96), oy = 64 + ((t * 2) % 64);
    for (int y = 0; y < e->H; y++) for (int x = 0; x < e->W; x++) e->src.pl[0][y * e->src.stride[0] + x] = e->tex[(y + oy) * e->tex_w + x + ox];
    for (int k = 0; k < 2; k++) {
        int bw = 24 + 16 * k, bx = (e->W > bw ? (17 * k + t * (5 - 3 * k)) % (e->W - bw) : 0), by = (e->H > bw ? (29 * k + t * (2 + k)) % (e->H - bw) : 0);
        for (int y = by; y < by + bw && y < e->H; y++) for (int x = bx; x < bx + bw && x < e->W; x++)
            e->src.pl[0][y * e->src.stride[0] + x] = (uint8_t)CLIP1(e->tex[(y - by + 8 * k) * e->tex_w + (x - bx) + 200 * k] + 30 - 60 * k);
    }
    for (int c = 1; c < 3; c++) for (int y = 0; y < e->H / 2; y++) for (int x = 0; x < e->W / 2; x++) {
        int v = e->tex[((y + oy / 2) * 2 + c * 37) % e->tex_h * e->tex_w + ((x + ox / 2) * 2 + c * 91) % e->tex_w];
        e->src.pl[c][y * e->src.stride[c] + x] = (uint8_t)(128 + (v - 128) / 3);
    }
}

/* ------------------------------ availability (6.4.1 / 6.4.2) ------------------------------ */
/* z-order rank of a 4x4 unit: CTB in tile scan first, then bit-interleaved position inside the CTB */
static uint32_t zrank(const Enc *e, int x, int y) {
    int cl = e->p.ctb_log2, rs = (y >> cl) * e->ctb_w + (x >> cl);
    uint32_t v = (uint32_t)e->rs2ts[rs] << 16;
    int lx = (x & (e->ctb - 1)) >> 2, ly = (y & (e->ctb - 1)) >> 2;
    for (int i = 0; i < 4; i++) v |= (uint32_t)(((lx >> i) & 1) << (2 * i)) | (uint32_t)(((ly >> i) & 1) << (2 * i + 1));
    return v;
}
static int avail(const Enc *e, int xc, int yc, int xn, int yn) {
    if (xn < 0 || yn < 0 || xn >= e->W || yn >= e->H) return 0;
    if (zrank(e, xn, yn) > zrank(e, xc, yc)) return 0;
    int cl = e->p.ctb_log2, cn = (yn >> cl) * e->ctb_w + (xn >> cl), cc = (yc >> cl) * e->ctb_w + (xc >> cl);
    return e->ctb_slice[cn] == e->seg_first_ctb && e->tile_of[e->rs2ts[cn]] == e->tile_of[e->rs2ts[cc]] && e->pm[I4(e, xn, yn)] != 0;
}
static int avail_pu(const Enc *e, int xcb, int ycb, int ncb, int xp, int yp, int w, int h, int part, int xn, int yn) {
    int inside = xn >= xcb && yn >= ycb && xn < xcb + ncb && yn < ycb + ncb, a;
    if (inside) a = !(2 * w == ncb && 2 * h == ncb && part == 1 && yn >= ycb + h && xn < xcb + w);
    else a = avail(e, xp, yp, xn, yn);
    return a && e->pm[I4(e, xn, yn)] == 1;
}

/* ------------------------------ intra prediction (8.4.4.2), generator's own statement ------------------------------ */
/* edge samples are gathered into one array going from the bottom-left corner up to the corner and then right along the top:
 *   r[0 .. 2n-1] = left column bottom-to-top, r[2n] = corner, r[2n+1 .. 4n] = top row left-to-right */
static void intra_edges(Enc *e, int x0, int y0, int n, int c, int *r) {
    Pic *p = e->cur; int sc = c ? 1 : 0, st = p->stride[c]; const uint8_t *pl = p->pl[c];
    uint8_t ok[4 * 32 + 1];
    int xl = x0 << sc, yl = y0 << sc, unit = c ? 2 : 4, cip = e->p.cip;
    for (int i = 0; i < 2 * n; i += unit) {
        int yy = y0 + 2 * n - 1 - i;                                    /* r index i <-> left sample row yy */
        int a = avail(e, xl, yl, xl - 1, yy << sc); if (a && cip && e->pm[I4(e, xl - 1, yy << sc)] != 2) a = 0;
        for (int k = 0; k < unit; k++) { ok[i + k] = (uint8_t)a; if (a) r[i + k] = pl[(y0 + 2 * n - 1 - i - k) * st + x0 - 1]; }
        int xx = x0 + i;
        a = avail(e, xl, yl, xx << sc, yl - 1); if (a && cip && e->pm[I4(e, xx << sc, yl - 1)] != 2) a = 0;
        for (int k = 0; k < unit; k++) { ok[2 * n + 1 + i + k] = (uint8_t)a; if (a) r[2 * n + 1 + i + k] = pl[(y0 - 1) * st + x0 + i + k]; }
    }
    { int a = avail(e, xl, yl, xl - 1, yl - 1); if (a && cip && e->pm[I4(e, xl - 1, yl - 1)] != 2) a = 0; ok[2 * n] = (uint8_t)a;
        if (a) r[2 * n] = pl[(y0 - 1) * st + x0 - 1]; }
    int first = -1;
    for (int i = 0; i <= 4 * n; i++) if (ok[i]) { first = i; break; }
    if (first < 0) { for (int i = 0; i <= 4 * n; i++) r[i] = 128; return; }
    for (int i = 0; i < first; i++) r[i] = r[first];
    for (int i = first + 1; i <= 4 * n; i++) if (!ok[i]) r[i] = r[i - 1];
}
static void intra_predict(Enc *e, int x0, int y0, int log2, int c, int mode, uint8_t *dst, int dstride) {
    int n = 1 << log2, r_[4 * 32 + 1], f_[4 * 32 + 1], *r = r_;
    intra_edges(e, x0, y0, n, c, r);
    if (c == 0 && mode != 1 && n > 4) {
        int dv = ABS(mode - 26), dh = ABS(mode - 10), md = MIN(dv, dh), thr = n == 8 ? 7 : (n == 16 ? 1 : 0);
        if (md > thr) {
            int N = 4 * n;
            if (e->p.strong_intra && n == 32 && ABS(r[64] + r[128] - 2 * r[96]) < 8 && ABS(r[64] + r[0] - 2 * r[32]) < 8) {
                f_[64] = r[64]; f_[0] = r[0]; f_[128] = r[128];
                for (int i = 1; i < 64; i++) { f_[64 - i] = ((64 - i) * r[64] + i * r[0] + 32) >> 6; f_[64 + i] = ((64 - i) * r[64] + i * r[128] + 32) >> 6; }
            } else { f_[0] = r[0]; f_[N] = r[N]; for (int i = 1; i < N; i++) f_[i] = (r[i - 1] + 2 * r[i] + r[i + 1] + 2) >> 2; }
            r = f_;
        }
    }
    const int *L = r + 2 * n - 1, *T = r + 2 * n + 1;                  /* L[-y] = left sample of row y; T[x] = top sample of column x; T[-1] = L[1] = corner */
#define LEFT(y) L[-(y)]
#define TOP(x) T[(x)]
    if (mode == 0) {
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++)
            dst[y * dstride + x] = (uint8_t)(((n - 1 - x) * LEFT(y) + (x + 1) * TOP(n) + (n - 1 - y) * TOP(x) + (y + 1) * LEFT(n) + n) >> (log2 + 1));
    } else if (mode == 1) {
        int s = n; for (int i = 0; i < n; i++) s += LEFT(i) + TOP(i);
        int dc = s >> (log2 + 1);
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) dst[y * dstride + x] = (uint8_t)dc;
        if (c == 0 && n < 32) {
            dst[0] = (uint8_t)((LEFT(0) + 2 * dc + TOP(0) + 2) >> 2);
            for (int i = 1; i < n; i++) { dst[i] = (uint8_t)((TOP(i) + 3 * dc + 2) >> 2); dst[i * dstride] = (uint8_t)((LEFT(i) + 3 * dc + 2) >> 2); }
        }
    } else {
        int ang = hg_intra_angle[mode], inv = hg_inv_angle[mode], vert = mode >= 18;
        int ref_[160], *ref = ref_ + 64;
        /* main reference = top row for vertical modes, left column for horizontal ones; index 0 = corner */
        for (int i = 0; i <= 2 * n; i++) ref[i] = vert ? TOP(i - 1) : LEFT(i - 1);
        if (ang < 0) for (int i = -1; i >= (n * ang) >> 5; i--) { int k = (i * inv + 128) >> 8; ref[i] = vert ? LEFT(k - 1) : TOP(k - 1); }
        for (int a = 0; a < n; a++) {                                  /* a runs along the prediction direction's minor axis */
            int pos = (a + 1) * ang, idx = pos >> 5, fr = pos & 31;
            for (int b = 0; b < n; b++) {
                int v = fr ? ((32 - fr) * ref[b + idx + 1] + fr * ref[b + idx + 2] + 16) >> 5 : ref[b + idx + 1];
                if (vert) dst[a * dstride + b] = (uint8_t)v; else dst[b * dstride + a] = (uint8_t)v;
            }
        }
        if (c == 0 && n < 32 && ang == 0) for (int i = 0; i < n; i++) {
            if (vert) dst[i * dstride] = (uint8_t)CLIP1(TOP(0) + ((LEFT(i) - TOP(-1)) >> 1)); else dst[i] = (uint8_t)CLIP1(LEFT(0) + ((TOP(i) - TOP(-1)) >> 1));
        }
    }
#undef LEFT
#undef TOP
}

/* ------------------------------ motion compensation (8.5.3.3.3), generator's own statement ------------------------------ */
static int tap_h(const uint8_t *pl, int st, int pw, int ph, int x, int y, const int8_t *f, int nt) {
    int v = 0, yy = CLIP3(0, ph - 1, y);
    for (int i = 0; i < nt; i++) v += f[i] * pl[yy * st + CLIP3(0, pw - 1, x + i - (nt / 2 - 1))];
    return v;
}
static void mc_pred(const Pic *ref, int c, int pw, int ph, int xb, int yb, int bw, int bh, int mvx, int mvy, int16_t *out) {
    int nt = c ? 4 : 8, sh = c ? 3 : 2, mask = (1 << sh) - 1;
    int fx = mvx & mask, fy = mvy & mask, xi = xb + (mvx >> sh), yi = yb + (mvy >> sh);
    const int8_t *hf = c ? hg_chroma_filter[fx] : hg_luma_filter[fx], *vf = c ? hg_chroma_filter[fy] : hg_luma_filter[fy];
    const uint8_t *pl = ref->pl[c]; int st = ref->stride[c];
    for (int y = 0; y < bh; y++) for (int x = 0; x < bw; x++) {
        int v;
        if (!fx && !fy) v = pl[CLIP3(0, ph - 1, yi + y) * st + CLIP3(0, pw - 1, xi + x)] << 6;
        else if (!fy) v = tap_h(pl, st, pw, ph, xi + x, yi + y, hf, nt);
        else if (!fx) { v = 0; for (int j = 0; j < nt; j++) v += vf[j] * pl[CLIP3(0, ph - 1, yi + y + j - (nt / 2 - 1)) * st + CLIP3(0, pw - 1, xi + x)]; }
        else { v = 0; for (int j = 0; j < nt; j++) v += vf[j] * tap_h(pl, st, pw, ph, xi + x, yi + y + j - (nt / 2 - 1), hf, nt); v >>= 6; }
        out[y * bw + x] = (int16_t)v;
    }
}
/* prediction of one block of plane c into dst (8.5.3.3.4) */
static void inter_block(Enc *e, const Mot *m, int c, int xp, int yp, int w, int h, uint8_t *dst, int dstride) {
    static __thread int16_t a[64 * 64], b[64 * 64];
    int sc = c ? 1 : 0, pw = e->W >> sc, ph = e->H >> sc, bw = w >> sc, bh = h >> sc;
    const Slc *s = e->sl;
    if (m->pf & 1) mc_pred(s->ref[0][m->ref[0]], c, pw, ph, xp >> sc, yp >> sc, bw, bh, m->mv[0][0], m->mv[0][1], a);
    if (m->pf & 2) mc_pred(s->ref[1][m->ref[1]], c, pw, ph, xp >> sc, yp >> sc, bw, bh, m->mv[1][0], m->mv[1][1], (m->pf & 1) ? b : a);
    for (int y = 0; y < bh; y++) for (int x = 0; x < bw; x++) {
        int i = y * bw + x, v;
        if (!s->wp_on) v = m->pf == 3 ? (a[i] + b[i] + 64) >> 7 : (a[i] + 32) >> 6;
        else {
            int ld = s->wp_denom[c ? 1 : 0] + 6;
            if (m->pf == 3)
                v = (a[i] * s->wp_w[0][m->ref[0]][c] + b[i] * s->wp_w[1][m->ref[1]][c] + ((s->wp_o[0][m->ref[0]][c] + s->wp_o[1][m->ref[1]][c] + 1) << ld)) >> (ld + 1);
            else { int l = m->pf == 1 ? 0 : 1; v = ((a[i] * s->wp_w[l][m->ref[l]][c] + (1 << (ld - 1))) >> ld) + s->wp_o[l][m->ref[l]][c]; }
        }
        dst[y * dstride + x] = (uint8_t)CLIP1(v);
    }
}

/* ------------------------------ motion candidates (8.5.3.2), generator's own statement ------------------------------ */
static int scale_mv(int mv, int td, int tb) {
    td = CLIP3(-128, 127, td); tb = CLIP3(-128, 127, tb);
    int tx = (16384 + ABS(td) / 2) / td, dsf = CLIP3(-4096, 4095, (tb * tx + 32) >> 6), p = dsf * mv;
    return CLIP3(-32768, 32767, SIGN(p) * ((ABS(p) + 127) >> 8));
}
static int same_mot(const Mot *a, const Mot *b) {
    if (a->pf != b->pf) return 0;
    for (int l = 0; l < 2; l++) if ((a->pf >> l) & 1) if (a->ref[l] != b->ref[l] || a->mv[l][0] != b->mv[l][0] || a->mv[l][1] != b->mv[l][1]) return 0;
    return 1;
}
static int temporal_cand(Enc *e, int xp, int yp, int w, int h, int X, int ridx, int16_t *mv) {
    const Slc *s = e->sl;
    if (!s->tmvp) return 0;
    const Pic *col = s->ref[(s->type == 0 && !s->col_l0) ? 1 : 0][s->col_idx];
    int cw = (e->W + 15) >> 4;
    for (int pass = 0; pass < 2; pass++) {
        int xc = pass ? xp + w / 2 : xp + w, yc = pass ? yp + h / 2 : yp + h;
        if (!pass && ((yp >> e->p.ctb_log2) != (yc >> e->p.ctb_log2) || xc >= e->W || yc >= e->H)) continue;
        int ce = (yc >> 4) * cw + (xc >> 4);
        if (col->col_intra[ce]) continue;
        const Mot *cm = &col->col[ce];
        int l;
        if (!(cm->pf & 1)) l = 1; else if (!(cm->pf & 2)) l = 0;
        else { int nb = 1; for (int k = 0; k < 2; k++) for (int i = 0; i < s->n_ref[k]; i++) if (s->ref_poc[k][i] > e->cur->poc) nb = 0;
            l = nb ? X : s->col_l0; }
        int lt = (col->col_lt[ce] >> l) & 1;
        if (lt != s->ref_lt[X][ridx]) continue;
        int cd = col->poc - col->col_poc[ce * 2 + l], bd = e->cur->poc - s->ref_poc[X][ridx];
        if (lt || cd == bd || cd == 0) { mv[0] = cm->mv[l][0]; mv[1] = cm->mv[l][1]; }
        else { mv[0] = (int16_t)scale_mv(cm->mv[l][0], cd, bd); mv[1] = (int16_t)scale_mv(cm->mv[l][1], cd, bd); }
        return 1;
    }
    return 0;
}
/* merge candidate list, up to `want` + 1 entries; returns the count */
static int merge_list(Enc *e, int xcb, int ycb, int ncb, int xp, int yp, int w, int h, int part, int part_mode, Mot *list) {
    const Slc *s = e->sl; int n = 0, pl = e->p.par_mrg;
    if (pl > 2 && ncb == 8) { xp = xcb; yp = ycb; w = h = 8; part = 0; part_mode = 0; }
    int nx[5] = { xp - 1, xp + w - 1, xp + w, xp - 1, xp - 1 }, ny[5] = { yp + h - 1, yp - 1, yp - 1, yp + h, yp - 1 };   /* A1 B1 B0 A0 B2 */
    int have[5]; Mot c[5];
    for (int k = 0; k < 5; k++) {
        have[k] = !((xp >> pl) == (nx[k] >> pl) && (yp >> pl) == (ny[k] >> pl)) && avail_pu(e, xcb, ycb, ncb, xp, yp, w, h, part, nx[k], ny[k]);
        if (k == 0 && part == 1 && (part_mode == 2 || part_mode == 6 || part_mode == 7)) have[k] = 0;      /* Nx2N nLx2N nRx2N */
        if (k == 1 && part == 1 && (part_mode == 1 || part_mode == 4 || part_mode == 5)) have[k] = 0;      /* 2NxN 2NxnU 2NxnD */
        if (have[k]) c[k] = e->mot[I4(e, nx[k], ny[k])];
    }
    /* 8.5.3.2.3: B0 / B2 are compared with B1 when B1 is available (availableB1), whether or not B1 itself was pruned against A1 */
    const int avail_b1 = have[1];
    if (have[1] && have[0] && same_mot(&c[1], &c[0])) have[1] = 0;
    if (have[2] && avail_b1 && same_mot(&c[2], &c[1])) have[2] = 0;
    if (have[3] && have[0] && same_mot(&c[3], &c[0])) have[3] = 0;
    if (have[4] && ((have[0] && same_mot(&c[4], &c[0])) || (avail_b1 && same_mot(&c[4], &c[1])) || have[0] + have[1] + have[2] + have[3] == 4)) have[4] = 0;
    for (int k = 0; k < 5; k++) if (have[k]) list[n++] = c[k];
    if (n < s->max_merge && s->tmvp) {
        Mot t; memset(&t, 0, sizeof t); t.ref[0] = t.ref[1] = -1;
        if (temporal_cand(e, xp, yp, w, h, 0, 0, t.mv[0])) { t.pf |= 1; t.ref[0] = 0; }
        if (s->type == 0 && temporal_cand(e, xp, yp, w, h, 1, 0, t.mv[1])) { t.pf |= 2; t.ref[1] = 0; }
        if (t.pf) list[n++] = t;
    }
    if (n > s->max_merge) n = s->max_merge;
    if (s->type == 0 && n > 1 && n < s->max_merge) {
        static const int a0[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, a1[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
        int no = n;
        for (int k = 0; k < no * (no - 1) && n < s->max_merge; k++) {
            const Mot *p = &list[a0[k]], *q = &list[a1[k]];
            if (!(p->pf & 1) || !(q->pf & 2)) continue;
            if (s->ref[0][p->ref[0]] == s->ref[1][q->ref[1]] && p->mv[0][0] == q->mv[1][0] && p->mv[0][1] == q->mv[1][1]) continue;
            Mot t; t.pf = 3; t.ref[0] = p->ref[0]; t.ref[1] = q->ref[1]; t.mv[0][0] = p->mv[0][0]; t.mv[0][1] = p->mv[0][1]; t.mv[1][0] = q->mv[1][0];
            t.mv[1][1] = q->mv[1][1];
            list[n++] = t;
        }
    }
    int nr = s->type == 1 ? s->n_ref[0] : MIN(s->n_ref[0], s->n_ref[1]);
    for (int z = 0; n < s->max_merge; z++) { Mot t; memset(&t, 0, sizeof t); t.pf = s->type == 1 ? 1 : 3; t.ref[0] = (int8_t)(z < nr ? z : 0);
        t.ref[1] = (int8_t)(s->type == 1 ? -1 : (z < nr ? z : 0)); list[n++] = t; }
    return n;
}
static void merge_fixup(Mot *m, int w, int h) {                        /* 8x4 / 4x8 blocks are never bi-predicted */
    if (m->pf == 3 && w + h == 12) { m->pf = 1; }
    for (int l = 0; l < 2; l++) if (!((m->pf >> l) & 1)) { m->ref[l] = -1; m->mv[l][0] = m->mv[l][1] = 0; }
}
/* the two AMVP candidates of list X / reference ridx */
static void amvp_list(Enc *e, int xcb, int ycb, int ncb, int xp, int yp, int w, int h, int part, int X, int ridx, int16_t out[2][2]) {
    const Slc *s = e->sl;
    int tp = s->ref_poc[X][ridx], tl = s->ref_lt[X][ridx], cur = e->cur->poc;
    int ax[2] = { xp - 1, xp - 1 }, ay[2] = { yp + h, yp + h - 1 }, bx[3] = { xp + w, xp + w - 1, xp - 1 }, by[3] = { yp - 1, yp - 1, yp - 1 };
    int okA[2], okB[3], gotA = 0, gotB = 0; int16_t A[2] = {0, 0}, B[2] = {0, 0};
    for (int k = 0; k < 2; k++) okA[k] = avail_pu(e, xcb, ycb, ncb, xp, yp, w, h, part, ax[k], ay[k]);
    for (int k = 0; k < 3; k++) okB[k] = avail_pu(e, xcb, ycb, ncb, xp, yp, w, h, part, bx[k], by[k]);
    for (int scaled = 0; scaled < 2 && !gotA; scaled++) for (int k = 0; k < 2 && !gotA; k++) if (okA[k]) {
        const Mot *m = &e->mot[I4(e, ax[k], ay[k])]; const Slc *ns = &e->slices[e->slice_of[I4(e, ax[k], ay[k])]];
        for (int t = 0; t < 2 && !gotA; t++) { int l = t ? !X : X; if (!((m->pf >> l) & 1)) continue;
            int rp = ns->ref_poc[l][m->ref[l]], rl = ns->ref_lt[l][m->ref[l]];
            if (!scaled ? rp == tp : rl == tl) { gotA = 1; A[0] = m->mv[l][0]; A[1] = m->mv[l][1];
                if (scaled && !rl && !tl && cur - rp != cur - tp && cur != rp) { A[0] = (int16_t)scale_mv(A[0], cur - rp, cur - tp);
                    A[1] = (int16_t)scale_mv(A[1], cur - rp, cur - tp); } } }
    }
    for (int k = 0; k < 3 && !gotB; k++) if (okB[k]) {
        const Mot *m = &e->mot[I4(e, bx[k], by[k])]; const Slc *ns = &e->slices[e->slice_of[I4(e, bx[k], by[k])]];
        for (int t = 0; t < 2 && !gotB; t++) { int l = t ? !X : X; if (((m->pf >> l) & 1) && ns->ref_poc[l][m->ref[l]] == tp) { gotB = 1; B[0] = m->mv[l][0];
            B[1] = m->mv[l][1]; } }
    }
    if (!okA[0] && !okA[1]) {
        if (gotB) { gotA = 1; A[0] = B[0]; A[1] = B[1]; }
        gotB = 0;
        for (int k = 0; k < 3 && !gotB; k++) if (okB[k]) {
            const Mot *m = &e->mot[I4(e, bx[k], by[k])]; const Slc *ns = &e->slices[e->slice_of[I4(e, bx[k], by[k])]];
            for (int t = 0; t < 2 && !gotB; t++) { int l = t ? !X : X; if (!((m->pf >> l) & 1)) continue;
                int rp = ns->ref_poc[l][m->ref[l]], rl = ns->ref_lt[l][m->ref[l]];
                if (rl == tl) { gotB = 1; B[0] = m->mv[l][0]; B[1] = m->mv[l][1];
                    if (!rl && !tl && rp != tp && cur != rp) { B[0] = (int16_t)scale_mv(B[0], cur - rp, cur - tp);
                        B[1] = (int16_t)scale_mv(B[1], cur - rp, cur - tp); } } }
        }
    }
    int n = 0;
    if (gotA) { out[n][0] = A[0]; out[n][1] = A[1]; n++; }
    if (gotB && !(gotA && A[0] == B[0] && A[1] == B[1])) { out[n][0] = B[0]; out[n][1] = B[1]; n++; }
    if (n < 2) { int16_t t[2]; if (temporal_cand(e, xp, yp, w, h, X, ridx, t)) { out[n][0] = t[0]; out[n][1] = t[1]; n++; } }
    for (; n < 2; n++) out[n][0] = out[n][1] = 0;
}

/* ------------------------------ transforms and quantisation ------------------------------ */
static int g_basis_ready; static int16_t g_dct[6][32][32];           /* g_dct[log2][k][n]: row k of the nTbS-point core transform */
static void basis_init(void) {
    if (g_basis_ready) return;
    for (int l = 2; l <= 5; l++) { int n = 1 << l; for (int k = 0; k < n; k++) for (int i = 0; i < n; i++) g_dct[l][k][i] = hg_trans[k * (32 >> l)][i]; }
    g_basis_ready = 1;
}
/* forward transform (encoder side, not normative): coefficient k = sum_n basis[k][n] * x[n], scaled like HM's */
static void fwd_transform(const int *res, int *coef, int log2, int dst) {
    int n = 1 << log2, tmp[32 * 32];
    int s1 = log2 + 8 - 9 + 0, s2 = log2 + 6;                         /* first stage shift = log2 - 1, second = log2 + 6 (8-bit) */
    for (int y = 0; y < n; y++) for (int k = 0; k < n; k++) {          /* rows */
        int64_t v = 0; for (int x = 0; x < n; x++) v += (int64_t)(dst ? hg_dst[k][x] : g_dct[log2][k][x]) * res[y * n + x];
        tmp[y * n + k] = (int)((v + (s1 > 0 ? (1 << (s1 - 1)) : 0)) >> s1);
    }
    for (int x = 0; x < n; x++) for (int k = 0; k < n; k++) {          /* columns */
        int64_t v = 0; for (int y = 0; y < n; y++) v += (int64_t)(dst ? hg_dst[k][y] : g_dct[log2][k][y]) * tmp[y * n + x];
        coef[k * n + x] = (int)((v + (1 << (s2 - 1))) >> s2);
    }
}
/* normative reconstruction of the residual from levels (8.6.2 - 8.6.4) */
static void inv_residual(const Enc *e, const int16_t *lev, int *res, int log2, int c, int intra, int tskip, int bypass, int qp) {
    int n = 1 << log2;
    if (bypass) { for (int i = 0; i < n * n; i++) res[i] = lev[i]; return; }
    int d[32 * 32], shift = log2 + 3, scale = hg_level_scale[qp % 6] << (qp / 6);
    const uint8_t *m = e->sf[log2 - 2][log2 == 5 ? (intra ? 0 : 1) : (intra ? 0 : 3) + c];
    int flat = !e->sf_on || (tskip && n > 4);
    for (int i = 0; i < n * n; i++) d[i] = lev[i] ? CLIP3(-32768, 32767,
        (int)(((int64_t)lev[i] * (flat ? 16 : m[i]) * scale + (1 << (shift - 1))) >> shift)) : 0;
    if (tskip) { for (int i = 0; i < n * n; i++) res[i] = ((d[i] << 7) + 2048) >> 12; return; }
    int dst = intra && c == 0 && n == 4, g[32 * 32];
    for (int x = 0; x < n; x++) for (int y = 0; y < n; y++) {          /* vertical stage: sample y of column x */
        int v = 0; for (int k = 0; k < n; k++) { int dk = d[k * n + x]; if (dk) v += (dst ? hg_dst[k][y] : g_dct[log2][k][y]) * dk; }
        g[y * n + x] = CLIP3(-32768, 32767, (v + 64) >> 7);
    }
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
        int v = 0; for (int k = 0; k < n; k++) v += (dst ? hg_dst[k][x] : g_dct[log2][k][x]) * g[y * n + k];
        res[y * n + x] = (v + 2048) >> 12;
    }
}
static void quantise(const Enc *e, const int *coef, int16_t *lev, int log2, int c, int intra, int qp, int tskip_src) {
    static const int qs[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };
    int n = 1 << log2, qbits = 14 + qp / 6 + (15 - 8 - log2), add = (intra ? 171 : 85) << (qbits - 9);
    const uint8_t *m = e->sf[log2 - 2][log2 == 5 ? (intra ? 0 : 1) : (intra ? 0 : 3) + c];
    for (int i = 0; i < n * n; i++) {
        int64_t a = (int64_t)ABS(coef[i]) * qs[qp % 6];
        if (e->sf_on && !tskip_src) a = a * 16 / m[i];
        int l = (int)((a + add) >> qbits);
        if (l > 2000) l = 2000;
        lev[i] = (int16_t)(coef[i] < 0 ? -l : l);
    }
}

/* ------------------------------ residual_coding writer (7.3.8.11 / 9.3.4.2) ------------------------------ */
static void scan_xy(int idx, int log2, int i, int *x, int *y) {
    int n = 1 << log2;
    if (idx == 1) { *x = i & (n - 1); *y = i >> log2; return; }
    if (idx == 2) { *x = i >> log2; *y = i & (n - 1); return; }
    int k = 0;
    for (int s = 0; s < 2 * n - 1; s++) for (int xx = 0; xx <= s; xx++) { int yy = s - xx; if (xx < n && yy < n) { if (k == i) { *x = xx; *y = yy; return; }
        k++; } }
}
static void put_last_prefix(Cab *cb, int base, int log2, int c, int v) {
    int cmax = 2 * log2 - 1, off = c ? 15 : 3 * (log2 - 2) + ((log2 - 1) >> 2), sh = c ? log2 - 2 : (log2 + 1) >> 2;
    for (int i = 0; i < v; i++) cab_enc(cb, base + off + (i >> sh), 1);
    if (v < cmax) cab_enc(cb, base + off + (v >> sh), 0);
}
static void last_split(int pos, int *prefix, int *suffix, int *nbits) {
    if (pos < 4) { *prefix = pos; *nbits = 0; *suffix = 0; return; }
    int p = 4;
    for (;; p++) { int nb = (p >> 1) - 1, base = (1 << nb) * (2 + (p & 1)); if (pos >= base && pos < base + (1 << nb)) { *prefix = p; *nbits = nb;
        *suffix = pos - base; return; } }
}
/* make the levels of one transform block expressible and (with sign hiding) parity-consistent; returns 1 when any level is non-zero */
static int shape_levels(const Enc *e, int16_t *lev, int log2, int scan, int bypass) {
    int n = 1 << log2, any = 0, nsb = 1 << (2 * (log2 - 2));
    for (int i = 0; i < n * n; i++) any |= lev[i] != 0;
    if (!any || !e->p.sdh || bypass) return any;
    for (int sb = 0; sb < nsb; sb++) {
        int xs, ys, first = -1, last = -1, sum = 0, xp, yp;
        scan_xy(scan, log2 - 2, sb, &xs, &ys);
        for (int k = 0; k < 16; k++) { scan_xy(scan, 2, k, &xp, &yp); int v = lev[((ys << 2) + yp) * n + (xs << 2) + xp]; if (v) { if (first < 0) first = k;
            last = k; sum += ABS(v); } }
        if (first < 0 || last - first <= 3) continue;
        scan_xy(scan, 2, first, &xp, &yp);                             /* lowest scan position = the coefficient whose sign is hidden */
        int16_t *f = &lev[((ys << 2) + yp) * n + (xs << 2) + xp];
        if ((sum & 1) != (*f < 0)) {                                    /* fix the parity on the last (highest frequency) coefficient of the group */
            scan_xy(scan, 2, last, &xp, &yp);
            int16_t *l = &lev[((ys << 2) + yp) * n + (xs << 2) + xp];
            *l = (int16_t)(*l + (*l < 0 ? -1 : 1));
        }
    }
    return 1;
}
static void write_residual(Enc *e, const int16_t *lev, int log2, int c, int scan, int tskip, int bypass) {
    Cab *cb = &e->cab; int n = 1 << log2, nsl = log2 - 2, nsb = 1 << nsl;
    if (e->p.tskip && !bypass && log2 == 2) cab_enc(cb, HG_CTX_TSKIP + (c ? 1 : 0), tskip);
    int last_sb = -1, last_pos = -1, xs, ys, xp, yp;
    for (int sb = nsb * nsb - 1; sb >= 0 && last_sb < 0; sb--) { scan_xy(scan, nsl, sb, &xs, &ys); for (int k = 15; k >= 0; k--) {
        scan_xy(scan, 2, k, &xp, &yp); if (lev[((ys << 2) + yp) * n + (xs << 2) + xp]) { last_sb = sb; last_pos = k; break; } } }
    scan_xy(scan, nsl, last_sb, &xs, &ys); scan_xy(scan, 2, last_pos, &xp, &yp);
    int lx = (xs << 2) + xp, ly = (ys << 2) + yp;
    if (scan == 2) { int t = lx; lx = ly; ly = t; }
    int px, sx, nx, py, sy, ny;
    last_split(lx, &px, &sx, &nx); last_split(ly, &py, &sy, &ny);
    put_last_prefix(cb, HG_CTX_LAST_X, log2, c, px); put_last_prefix(cb, HG_CTX_LAST_Y, log2, c, py);
    if (px > 3) cab_byp_n(cb, nx, (uint32_t)sx);
    if (py > 3) cab_byp_n(cb, ny, (uint32_t)sy);
    uint8_t coded[8][8]; memset(coded, 0, sizeof coded);
    int g1ctx = 1, first_group = 1;
    for (int sb = last_sb; sb >= 0; sb--) {
        scan_xy(scan, nsl, sb, &xs, &ys);
        int v16[16], nz = 0;
        for (int k = 0; k < 16; k++) { scan_xy(scan, 2, k, &xp, &yp); v16[k] = lev[((ys << 2) + yp) * n + (xs << 2) + xp]; nz += v16[k] != 0; }
        int right = xs < nsb - 1 ? coded[ys][xs + 1] : 0, below = ys < nsb - 1 ? coded[ys + 1][xs] : 0, infer_dc = 0;
        if (sb < last_sb && sb > 0) { coded[ys][xs] = nz != 0; cab_enc(cb, HG_CTX_CSBF + ((right | below) ? 1 : 0) + (c ? 2 : 0), nz != 0); infer_dc = 1; }
        else coded[ys][xs] = 1;
        if (!coded[ys][xs]) continue;
        int prev = right | (below << 1);
        for (int k = sb == last_sb ? last_pos - 1 : 15; k >= 0; k--) {
            if (k == 0 && infer_dc) break;                              /* DC of a coded group whose other 15 are zero is inferred */
            scan_xy(scan, 2, k, &xp, &yp);
            int xc = (xs << 2) + xp, yc = (ys << 2) + yp, sc;
            if (log2 == 2) { static const uint8_t map[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8}; sc = map[(yc << 2) + xc]; }
            else if (xc + yc == 0) sc = 0;
            else {
                sc = prev == 0 ? (xp + yp == 0 ? 2 : (xp + yp < 3 ? 1 : 0)) : prev == 1 ? (yp == 0 ? 2 : (yp == 1 ? 1 : 0)) : prev == 2 ?
                    (xp == 0 ? 2 : (xp == 1 ? 1 : 0)) : 2;
                if (c == 0) { if (xs + ys > 0) sc += 3; sc += log2 == 3 ? (scan == 0 ? 9 : 15) : 21; } else sc += log2 == 3 ? 9 : 12;
            }
            cab_enc(cb, HG_CTX_SIG + (c ? 27 + sc : sc), v16[k] != 0);
            if (v16[k]) infer_dc = 0;
        }
        if (!nz) continue;                                              /* cannot happen: a coded group has a non-zero level (the DC when inferred) */
        int pos[16], np = 0;
        for (int k = 15; k >= 0; k--) if (v16[k]) pos[np++] = k;
        int cset = (sb == 0 || c) ? 0 : 2;
        if (!first_group && g1ctx == 0) cset++;
        first_group = 0; g1ctx = 1;
        int lastg1 = -1;
        for (int m = 0; m < np && m < 8; m++) {
            int g = ABS(v16[pos[m]]) > 1;
            cab_enc(cb, HG_CTX_G1 + cset * 4 + g1ctx + (c ? 16 : 0), g);
            if (g) { g1ctx = 0; if (lastg1 < 0) lastg1 = m; } else if (g1ctx > 0 && g1ctx < 3) g1ctx++;
        }
        if (lastg1 >= 0) cab_enc(cb, HG_CTX_G2 + cset + (c ? 4 : 0), ABS(v16[pos[lastg1]]) > 2);
        int hide = e->p.sdh && !bypass && pos[0] - pos[np - 1] > 3;
        for (int m = 0; m < np - (hide ? 1 : 0); m++) cab_byp(cb, v16[pos[m]] < 0);
        int rice = 0;
        for (int m = 0; m < np; m++) {
            int a = ABS(v16[pos[m]]), base = 1 + (m < 8 ? (a > 1) : 0) + (m == lastg1 ? (a > 2) : 0);
            int thr = m < 8 ? (m == lastg1 ? 3 : 2) : 1;
            if (base == thr) {
                int rem = a - base, q = rem >> rice;
                if (q < 4) { for (int i = 0; i < q; i++) cab_byp(cb, 1); cab_byp(cb, 0); cab_byp_n(cb, rice, (uint32_t)(rem & ((1 << rice) - 1))); }
                else { for (int i = 0; i < 4; i++) cab_byp(cb, 1); cab_egk(cb, rice + 1, (uint32_t)(rem - (4 << rice))); }
                if (a > 3 * (1 << rice)) rice = MIN(rice + 1, 4);
            }
        }
    }
}

/* ------------------------------ coding unit: decisions + reconstruction, then syntax ------------------------------ */
typedef struct { int x, y, log2, depth, blk, split, cbf[3], tskip[3], child[4], xb, yb; int16_t *lev[3]; } Tu;
typedef struct { int x, y, w, h, merge, merge_idx, idc /* 0 L0 1 L1 2 BI */, ref[2], mvp[2]; int16_t mvd[2][2]; Mot m; } Pu;
typedef struct {
    int x, y, log2, skip, intra, part, bypass, pcm, n_pu, ipm[4], prev_flag[4], mpm_idx[4], rem[4], chroma_idx, ipm_c, root_cbf, qp, root;
    Pu pu[4];
    Tu tu[400]; int n_tu;
    int16_t lev[64 * 64 * 2]; int n_lev;
    int max_depth, intra_split;
} Cu;
static __thread Cu g_cu;            /* per thread: bench.py generates streams on several threads */

static int chroma_qp_of(const Enc *e, int qp, int c) { return hg_qpc_tab[CLIP3(0, 57, qp + (c == 1 ? e->p.cb_qp_off : e->p.cr_qp_off))]; }
static int scan_of(const Enc *e, const Cu *cu, int x0, int y0, int log2, int c) {
    if (!cu->intra || !(log2 == 2 || (log2 == 3 && c == 0))) return 0;
    int pm = c == 0 ? e->ipm[I4(e, x0, y0)] : cu->ipm_c;
    return pm >= 6 && pm <= 14 ? 2 : (pm >= 22 && pm <= 30 ? 1 : 0);
}
/* one transform block: (intra) predict, residual, transform + quantise, reconstruct.  x0/y0 in samples of plane c. */
static int code_tb(Enc *e, Cu *cu, Tu *t, int x0, int y0, int log2, int c, int xl, int yl) {
    int n = 1 << log2, st = e->cur->stride[c], res[32 * 32], coef[32 * 32];
    uint8_t *rec = e->cur->pl[c] + y0 * st + x0; const uint8_t *src = e->src.pl[c] + y0 * e->src.stride[c] + x0;
    if (cu->intra) intra_predict(e, x0, y0, log2, c, c ? cu->ipm_c : e->ipm[I4(e, xl, yl)], rec, st);
    int16_t *lev = cu->lev + cu->n_lev; cu->n_lev += n * n; t->lev[c] = lev;
    int qp = c ? chroma_qp_of(e, cu->qp, c) : cu->qp;
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) res[y * n + x] = src[y * e->src.stride[c] + x] - rec[y * st + x];
    int tskip = 0;
    if (cu->bypass) { for (int i = 0; i < n * n; i++) lev[i] = (int16_t)res[i]; }
    else {
        if (e->p.tskip && log2 == 2 && rnd_n(&e->rng, 4) == 0) tskip = 1;
        if (tskip) for (int i = 0; i < n * n; i++) coef[i] = res[i] << 5; else fwd_transform(res, coef, log2, cu->intra && c == 0 && n == 4);
        quantise(e, coef, lev, log2, c, cu->intra, qp, tskip);
        if (e->p.mode == 1) {                                           /* fuzz: sprinkle / clear levels so every residual syntax path shows up */
            int r = rnd_n(&e->rng, 8);
            if (r == 0) memset(lev, 0, sizeof(int16_t) * (size_t)(n * n));
            else if (r == 1) for (int k = 0; k < 3; k++) lev[rnd_n(&e->rng, n * n)] = (int16_t)(rnd_n(&e->rng, 41) - 20);
            else if (r == 2) lev[rnd_n(&e->rng, n * n)] = (int16_t)(rnd_n(&e->rng, 2) ? 300 + rnd_n(&e->rng, 1500) : -(300 + rnd_n(&e->rng, 1500)));
        }
    }
    t->tskip[c] = tskip;
    int any = shape_levels(e, lev, log2, scan_of(e, cu, xl, yl, log2, c), cu->bypass);
    if (any) {
        inv_residual(e, lev, res, log2, c, cu->intra, tskip, cu->bypass, qp);
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) rec[y * st + x] = (uint8_t)CLIP1(rec[y * st + x] + res[y * n + x]);
    }
    return any;
}
static int decide_tt(Enc *e, Cu *cu, int x0, int y0, int xb, int yb, int log2, int depth, int blk) {
    int id = cu->n_tu++; Tu *t = &cu->tu[id];
    memset(t, 0, sizeof *t); t->x = x0; t->y = y0; t->log2 = log2; t->depth = depth; t->blk = blk; t->xb = xb; t->yb = yb;
    const HevcGenParams *p = &e->p;
    int inter_split = p->depth_inter == 0 && !cu->intra && cu->part != 0 && depth == 0;
    if (log2 <= p->max_tb_log2 && log2 > p->min_tb_log2 && depth < cu->max_depth && !(cu->intra_split && depth == 0)) t->split = rnd_n(&e->rng,
        p->mode ? 2 : 5) == 0;
    else t->split = log2 > p->max_tb_log2 || (cu->intra_split && depth == 0) || inter_split;
    if (t->split) {
        int h = 1 << (log2 - 1);
        for (int k = 0; k < 4; k++) { int ch = decide_tt(e, cu, x0 + (k & 1) * h, y0 + (k >> 1) * h, x0, y0, log2 - 1, depth + 1, k); t = &cu->tu[id];
            t->child[k] = ch; }
        for (int k = 0; k < 4; k++) { const Tu *c = &cu->tu[t->child[k]]; t->cbf[0] |= c->cbf[0]; if (log2 > 3) { t->cbf[1] |= c->cbf[1];
            t->cbf[2] |= c->cbf[2]; } }
        if (log2 == 3) { const Tu *c3 = &cu->tu[t->child[3]]; t->cbf[1] = c3->cbf[1]; t->cbf[2] = c3->cbf[2]; }
        return id;
    }
    int n = 1 << log2;
    for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) { if (x == x0) e->edges[I4(e, x, y)] |= 1;
        if (y == y0) e->edges[I4(e, x, y)] |= 2; }
    t->cbf[0] = code_tb(e, cu, t, x0, y0, log2, 0, x0, y0);
    for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) e->cbf[I4(e, x, y)] = (uint8_t)t->cbf[0];
    if (log2 > 2) for (int c = 1; c < 3; c++) t->cbf[c] = code_tb(e, cu, t, x0 >> 1, y0 >> 1, log2 - 1, c, x0, y0);
    else if (blk == 3) for (int c = 1; c < 3; c++) t->cbf[c] = code_tb(e, cu, t, xb >> 1, yb >> 1, 2, c, xb, yb);
    return id;
}
static void write_tt(Enc *e, Cu *cu, int id, int pcb, int pcr) {
    Tu *t = &cu->tu[id]; Cab *cb = &e->cab; const HevcGenParams *p = &e->p;
    int log2 = t->log2, depth = t->depth;
    if (log2 <= p->max_tb_log2 && log2 > p->min_tb_log2 && depth < cu->max_depth && !(cu->intra_split && depth == 0)) cab_enc(cb, HG_CTX_SPLIT_TF + 5 - log2,
        t->split);
    int ccb = pcb, ccr = pcr;
    if (log2 > 2) {
        /* a split 8x8 node carries the chroma of its four 4x4 children; deeper nodes the OR over their subtree */
        ccb = t->cbf[1]; ccr = t->cbf[2];
        if (pcb) cab_enc(cb, HG_CTX_CBF_CBCR + depth, ccb); else ccb = 0;
        if (pcr) cab_enc(cb, HG_CTX_CBF_CBCR + depth, ccr); else ccr = 0;
    }
    if (t->split) { for (int k = 0; k < 4; k++) write_tt(e, cu, t->child[k], ccb, ccr); return; }
    int cbf_y = t->cbf[0];
    if (cu->intra || depth != 0 || ccb || ccr) cab_enc(cb, HG_CTX_CBF_LUMA + (depth == 0), cbf_y);
    if (cbf_y || ccb || ccr) {
        if (p->dqp && !e->dqp_coded) {
            int v = e->dqp_val, a = ABS(v);
            cab_enc(cb, HG_CTX_CU_QP_DELTA, a > 0);
            if (a > 0) { for (int i = 1; i < MIN(a, 5); i++) cab_enc(cb, HG_CTX_CU_QP_DELTA + 1, 1); if (a < 5) cab_enc(cb, HG_CTX_CU_QP_DELTA + 1, 0);
                else cab_egk(cb, 0, (uint32_t)(a - 5)); cab_byp(cb, v < 0); }
            e->dqp_coded = 1;
        }
    }
    if (cbf_y) write_residual(e, t->lev[0], log2, 0, scan_of(e, cu, t->x, t->y, log2, 0), t->tskip[0], cu->bypass);
    if (log2 > 2) { for (int c = 1; c < 3; c++) if (c == 1 ? ccb : ccr) write_residual(e, t->lev[c], log2 - 1, c, scan_of(e, cu, t->x, t->y, log2 - 1, c),
        t->tskip[c], cu->bypass); }
    else if (t->blk == 3) { for (int c = 1; c < 3; c++) if (c == 1 ? ccb : ccr) write_residual(e, t->lev[c], 2, c, scan_of(e, cu, t->xb, t->yb, 2, c),
        t->tskip[c], cu->bypass); }
}

/* 8.6.1 prediction of the luma QP for the quantisation group that holds (xcb, ycb) */
static int qp_pred(const Enc *e, int xcb, int ycb) {
    int lq = e->p.ctb_log2 - (e->p.dqp ? e->p.dqp - 1 : 0), xq = xcb & ~((1 << lq) - 1), yq = ycb & ~((1 << lq) - 1);
    int prev = e->first_qg ? e->sl->qp : e->qp_prev, a = prev, b = prev, cm = ~(e->ctb - 1);
    if (!e->p.dqp) return e->sl->qp;
    if (avail(e, xcb, ycb, xq - 1, yq) && ((xq - 1) & cm) == (xq & cm)) a = e->qpmap[I4(e, xq - 1, yq)];
    if (avail(e, xcb, ycb, xq, yq - 1) && ((yq - 1) & cm) == (yq & cm)) b = e->qpmap[I4(e, xq, yq - 1)];
    return (a + b + 1) >> 1;
}

static int sad_block(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h) { int s = 0;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) s += ABS(a[y * as + x] - b[y * bs + x]); return s; }
static int sad_mot(Enc *e, const Mot *m, int x, int y, int w, int h) {
    static __thread uint8_t tmp[64 * 64];
    inter_block(e, m, 0, x, y, w, h, tmp, w);
    return sad_block(tmp, w, e->src.pl[0] + y * e->src.stride[0] + x, e->src.stride[0], w, h);
}
static void store_motion(Enc *e, const Mot *m, int x0, int y0, int w, int h) {
    for (int y = y0; y < y0 + h; y += 4) for (int x = x0; x < x0 + w; x += 4) { e->mot[I4(e, x, y)] = *m; if (x == x0) e->edges[I4(e, x, y)] |= 4;
        if (y == y0) e->edges[I4(e, x, y)] |= 8; }
}
/* choose the motion of one prediction unit and store it */
static void decide_pu(Enc *e, Cu *cu, Pu *pu, int part_idx, int force_merge) {
    const Slc *s = e->sl; Rng *r = &e->rng; int fuzz = e->p.mode == 1;
    Mot list[6]; int n = merge_list(e, cu->x, cu->y, 1 << cu->log2, pu->x, pu->y, pu->w, pu->h, part_idx, cu->part, list);
    int best_merge = 0, best_cost = 1 << 30;
    if (fuzz) best_merge = rnd_n(r, n);
    else for (int i = 0; i < n; i++) { Mot t = list[i]; merge_fixup(&t, pu->w, pu->h); int c = sad_mot(e, &t, pu->x, pu->y, pu->w, pu->h) + 8 * i;
        if (c < best_cost) { best_cost = c; best_merge = i; } }
    int use_merge = force_merge || (fuzz ? rnd_n(r, 2) : 0);
    Mot am; memset(&am, 0, sizeof am); am.ref[0] = am.ref[1] = -1;
    if (!force_merge) {
        /* AMVP: pick direction and references, then a motion vector near the predictor */
        int idc = 0;
        if (s->type == 0) { idc = rnd_n(r, 3); if (pu->w + pu->h == 12 && idc == 2) idc = rnd_n(r, 2); }
        pu->idc = idc;
        int amvp_cost = 0;
        for (int l = 0; l < 2; l++) {
            if (idc == (l ? 0 : 1)) continue;
            int ri = rnd_n(r, s->n_ref[l]); if (!fuzz && rnd_n(r, 4)) ri = 0;
            int16_t cand[2][2]; amvp_list(e, cu->x, cu->y, 1 << cu->log2, pu->x, pu->y, pu->w, pu->h, part_idx, l, ri, cand);
            pu->ref[l] = ri; am.ref[l] = (int8_t)ri; am.pf |= (uint8_t)(1 << l);
            int16_t mv[2];
            if (fuzz) { int k = rnd_n(r, 2); pu->mvp[l] = k; int big = rnd_n(r, 16) == 0; for (int d = 0; d < 2; d++) {
                int t = cand[k][d] + (big ? rnd_n(r, 1025) - 512 : rnd_n(r, 33) - 16); mv[d] = (int16_t)CLIP3(-2048, 2047, t); }; }
            else {
                /* true motion of the panning texture relative to the reference, plus a small search */
                int dpoc = e->cur->poc - s->ref_poc[l][ri], gx = 0, gy = 0;
                { int t0 = e->cur->poc, t1 = t0 - dpoc; gx = ((64 + (t1 * 3) % 96) - (64 + (t0 * 3) % 96)) * 4;
                    gy = ((64 + (t1 * 2) % 64) - (64 + (t0 * 2) % 64)) * 4; }
                int bestc = 1 << 30; int16_t bmv[2] = {0, 0};
                for (int k = 0; k < 6; k++) {
                    Mot t; memset(&t, 0, sizeof t); t.pf = (uint8_t)(1 << l); t.ref[l] = (int8_t)ri; t.ref[!l] = -1;
                    int16_t tv[2];
                    if (k == 0) { tv[0] = (int16_t)gx; tv[1] = (int16_t)gy; } else if (k == 1) { tv[0] = cand[0][0]; tv[1] = cand[0][1]; } else if (k == 2) {
                        tv[0] = cand[1][0]; tv[1] = cand[1][1]; }
                    else { tv[0] = (int16_t)(gx + rnd_n(r, 4 * e->p.search + 1) - 2 * e->p.search);
                        tv[1] = (int16_t)(gy + rnd_n(r, 4 * e->p.search + 1) - 2 * e->p.search); }
                    tv[0] = (int16_t)CLIP3(-2048, 2047, tv[0]); tv[1] = (int16_t)CLIP3(-2048, 2047, tv[1]);
                    t.mv[l][0] = tv[0]; t.mv[l][1] = tv[1];
                    int c = sad_mot(e, &t, pu->x, pu->y, pu->w, pu->h);
                    if (c < bestc) { bestc = c; bmv[0] = tv[0]; bmv[1] = tv[1]; }
                }
                mv[0] = bmv[0]; mv[1] = bmv[1]; amvp_cost += bestc;
                int d0 = ABS(mv[0] - cand[0][0]) + ABS(mv[1] - cand[0][1]), d1 = ABS(mv[0] - cand[1][0]) + ABS(mv[1] - cand[1][1]);
                pu->mvp[l] = d1 < d0;
            }
            if (l == 1 && s->mvd_l1_zero && idc == 2) { mv[0] = cand[pu->mvp[1]][0]; mv[1] = cand[pu->mvp[1]][1]; }
            am.mv[l][0] = mv[0]; am.mv[l][1] = mv[1];
            pu->mvd[l][0] = (int16_t)(mv[0] - cand[pu->mvp[l]][0]); pu->mvd[l][1] = (int16_t)(mv[1] - cand[pu->mvp[l]][1]);
        }
        if (!fuzz) { if (idc == 2) amvp_cost = sad_mot(e, &am, pu->x, pu->y, pu->w, pu->h); use_merge = best_cost <= amvp_cost + 16 || rnd_n(r, 8) == 0; }
    }
    pu->merge = use_merge; pu->merge_idx = best_merge;
    if (use_merge) { pu->m = list[best_merge]; merge_fixup(&pu->m, pu->w, pu->h); } else pu->m = am;
    store_motion(e, &pu->m, pu->x, pu->y, pu->w, pu->h);
}
static void write_pu(Enc *e, const Cu *cu, const Pu *pu) {
    Cab *cb = &e->cab; const Slc *s = e->sl;
    if (!cu->skip) cab_enc(cb, HG_CTX_MERGE_FLAG, pu->merge);
    if (pu->merge) {
        if (s->max_merge > 1) { cab_enc(cb, HG_CTX_MERGE_IDX, pu->merge_idx > 0);
            for (int i = 1; i < s->max_merge - 1 && i <= pu->merge_idx; i++) cab_byp(cb, pu->merge_idx > i); }
        return;
    }
    if (s->type == 0) {
        if (pu->w + pu->h != 12) { cab_enc(cb, HG_CTX_INTER_PRED_IDC + e->depth[I4(e, pu->x, pu->y)], pu->idc == 2); }
        if (pu->idc != 2) cab_enc(cb, HG_CTX_INTER_PRED_IDC + 4, pu->idc);
    }
    for (int l = 0; l < 2; l++) {
        if (pu->idc == (l ? 0 : 1)) continue;
        if (s->n_ref[l] > 1) { int cmax = s->n_ref[l] - 1; for (int i = 0; i < cmax && i <= pu->ref[l]; i++) { int b = pu->ref[l] > i;
            if (i < 2) cab_enc(cb, HG_CTX_REF_IDX + i, b); else cab_byp(cb, b); } }
        if (!(l == 1 && s->mvd_l1_zero && pu->idc == 2)) {
            int ax = ABS(pu->mvd[l][0]), ay = ABS(pu->mvd[l][1]);
            cab_enc(cb, HG_CTX_MVD_G0, ax > 0); cab_enc(cb, HG_CTX_MVD_G0, ay > 0);
            if (ax) cab_enc(cb, HG_CTX_MVD_G1, ax > 1);
            if (ay) cab_enc(cb, HG_CTX_MVD_G1, ay > 1);
            if (ax) { if (ax > 1) cab_egk(cb, 1, (uint32_t)(ax - 2)); cab_byp(cb, pu->mvd[l][0] < 0); }
            if (ay) { if (ay > 1) cab_egk(cb, 1, (uint32_t)(ay - 2)); cab_byp(cb, pu->mvd[l][1] < 0); }
        }
        cab_enc(cb, HG_CTX_MVP_FLAG, pu->mvp[l]);
    }
}

static void intra_mpm(const Enc *e, int xp, int yp, int cand[3]) {
    int a = 1, b = 1;
    if (avail(e, xp, yp, xp - 1, yp) && e->pm[I4(e, xp - 1, yp)] == 2) a = e->ipm[I4(e, xp - 1, yp)];
    if (avail(e, xp, yp, xp, yp - 1) && e->pm[I4(e, xp, yp - 1)] == 2 && ((yp - 1) >> e->p.ctb_log2) == (yp >> e->p.ctb_log2)) b = e->ipm[I4(e, xp, yp - 1)];
    if (a == b) { if (a < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; } else { cand[0] = a; cand[1] = 2 + ((a + 29) & 31); cand[2] = 2 + ((a - 1) & 31); } }
    else { cand[0] = a; cand[1] = b; cand[2] = (a && b) ? 0 : ((a != 1 && b != 1) ? 1 : 26); }
}

static void encode_cu(Enc *e, int x0, int y0, int log2) {
    Cu *cu = &g_cu; Cab *cb = &e->cab; const HevcGenParams *p = &e->p; const Slc *s = e->sl; Rng *r = &e->rng;
    int n = 1 << log2, fuzz = p->mode == 1;
    memset(cu, 0, offsetof(Cu, pu)); cu->x = x0; cu->y = y0; cu->log2 = log2; cu->n_tu = 0; cu->n_lev = 0; cu->intra_split = 0; cu->max_depth = 0;
    cu->bypass = p->bypass && rnd_n(r, 12) == 0;
    /* ---- decisions ---- */
    int pred = qp_pred(e, x0, y0);
    if (p->dqp && !e->dqp_coded) { e->dqp_val = rnd_n(r, 3) ? rnd_n(r, 7) - 3 : (fuzz ? rnd_n(r, 41) - 20 : 0); }
    cu->qp = p->dqp ? ((pred + e->dqp_val + 52) % 52) : s->qp;
    int intra = s->type == 2;
    if (!intra) {
        if (fuzz) intra = rnd_n(r, 5) == 0;
        else { int act = 0; const uint8_t *sp = e->src.pl[0] + y0 * e->src.stride[0] + x0;
            for (int y = 0; y < n; y += 2) for (int x = 0; x < n; x += 2) act += ABS(sp[y * e->src.stride[0] + x] - sp[y * e->src.stride[0] + x + 1]);
            intra = rnd_n(r, 24) == 0 || (act < n * n / 8 && rnd_n(r, 3) == 0); }
    }
    if (p->pcm == 3) intra = 1;                                         /* known-answer streams: every coding unit I_PCM with 8-bit samples */
    cu->intra = intra;
    for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) {
        int i = I4(e, x, y);
        e->pm[i] = (uint8_t)(intra ? 2 : 1); e->skip[i] = 0; e->nofilt[i] = (uint8_t)cu->bypass; e->slice_of[i] = (int16_t)(e->sl - e->slices); e->edges[i] = 0;
        e->cbf[i] = 0; e->ipm[i] = 1;
        memset(&e->mot[i], 0, sizeof(Mot)); e->mot[i].ref[0] = e->mot[i].ref[1] = -1;
        if (x == x0) e->edges[i] |= 5;
        if (y == y0) e->edges[i] |= 10;
    }
    if (intra) {
        cu->pcm = p->pcm && log2 <= MIN(5, p->ctb_log2) && (p->pcm == 3 || rnd_n(r, fuzz ? 10 : 40) == 0);
        cu->part = (!cu->pcm && log2 == p->min_cb_log2 && rnd_n(r, 3) == 0) ? 3 : 0;
        if (cu->pcm) {
            for (int c = 0; c < 3; c++) { int sc = c ? 1 : 0; for (int y = y0 >> sc; y < (y0 + n) >> sc; y++) for (int x = x0 >> sc; x < (x0 + n) >> sc; x++) {
                int bits = p->pcm == 3 ? 8 : (c ? 6 : 7);
                e->cur->pl[c][y * e->cur->stride[c] + x] = (uint8_t)((e->src.pl[c][y * e->src.stride[c] + x] >> (8 - bits)) << (8 - bits)); } }
            if (p->pcm == 1 || p->pcm == 3) for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) e->nofilt[I4(e, x, y)] = 1;
            /* pcm_loop_filter_disabled_flag */
        } else {
            int np = cu->part == 3 ? 2 : 1, pb = n / np;
            cu->intra_split = np == 2;
            for (int k = 0; k < np * np; k++) {
                int xp = x0 + (k & 1) * pb, yp = y0 + (k >> 1) * pb, cand[3], mode;
                intra_mpm(e, xp, yp, cand);
                if (fuzz || rnd_n(r, 4) == 0) mode = rnd_n(r, 35);
                else {                                                  /* SAD over a handful of modes predicted from the real neighbours */
                    static uint8_t tmp[64 * 64]; int tryl[7] = {0, 1, 26, 10, cand[0], 2 + rnd_n(r, 33), 2 + rnd_n(r, 33)}, best = 1 << 30; mode = 0;
                    int pl = MIN(pb, 32) == pb ? log2 - (np == 2) : 5;
                    if (pb <= 32) for (int t = 0; t < 7; t++) { intra_predict(e, xp, yp, pl, 0, tryl[t], tmp, pb);
                        int c = sad_block(tmp, pb, e->src.pl[0] + yp * e->src.stride[0] + xp, e->src.stride[0], pb, pb); if (c < best) { best = c;
                        mode = tryl[t]; } }
                    else mode = tryl[rnd_n(r, 7)];
                }
                cu->ipm[k] = mode;
                int mi = -1; for (int i = 0; i < 3; i++) if (cand[i] == mode) mi = i;
                cu->prev_flag[k] = mi >= 0; cu->mpm_idx[k] = mi;
                if (mi < 0) { int srt[3] = {cand[0], cand[1], cand[2]}, t; if (srt[0] > srt[1]) { t = srt[0]; srt[0] = srt[1]; srt[1] = t; }
                    if (srt[0] > srt[2]) { t = srt[0]; srt[0] = srt[2]; srt[2] = t; } if (srt[1] > srt[2]) { t = srt[1]; srt[1] = srt[2]; srt[2] = t; }
                    int rem = mode; for (int i = 2; i >= 0; i--) if (rem > srt[i]) rem--; cu->rem[k] = rem; }
                for (int y = yp; y < yp + pb; y += 4) for (int x = xp; x < xp + pb; x += 4) e->ipm[I4(e, x, y)] = (uint8_t)mode;
            }
            static const int ctab[4] = {0, 26, 10, 1};
            cu->chroma_idx = rnd_n(r, 3) ? 4 : rnd_n(r, 4);
            cu->ipm_c = cu->chroma_idx == 4 ? cu->ipm[0] : (ctab[cu->chroma_idx] == cu->ipm[0] ? 34 : ctab[cu->chroma_idx]);
        }
    } else {
        /* partitioning */
        int part = 0, rr = rnd_n(r, fuzz ? 4 : 8);
        if (rr == 0) part = 1 + rnd_n(r, 2);
        else if (rr == 1 && p->amp && log2 > p->min_cb_log2) part = 4 + rnd_n(r, 4);
        else if (rr == 2 && log2 == p->min_cb_log2 && log2 > 3) part = 3;
        if (log2 == 3 && part >= 3) part = 0;
        cu->part = part;
        int w[4], h[4], xs[4], ys[4], np = 2; xs[0] = x0; ys[0] = y0;
        switch (part) {
        case 0: np = 1; w[0] = h[0] = n; break;
        case 1: w[0] = w[1] = n; h[0] = h[1] = n / 2; xs[1] = x0; ys[1] = y0 + n / 2; break;
        case 2: w[0] = w[1] = n / 2; h[0] = h[1] = n; xs[1] = x0 + n / 2; ys[1] = y0; break;
        case 3: np = 4; for (int k = 0; k < 4; k++) { w[k] = h[k] = n / 2; xs[k] = x0 + (k & 1) * n / 2; ys[k] = y0 + (k >> 1) * n / 2; } break;
        case 4: w[0] = w[1] = n; h[0] = n / 4; h[1] = 3 * n / 4; xs[1] = x0; ys[1] = y0 + n / 4; break;
        case 5: w[0] = w[1] = n; h[0] = 3 * n / 4; h[1] = n / 4; xs[1] = x0; ys[1] = y0 + 3 * n / 4; break;
        case 6: h[0] = h[1] = n; w[0] = n / 4; w[1] = 3 * n / 4; xs[1] = x0 + n / 4; ys[1] = y0; break;
        default: h[0] = h[1] = n; w[0] = 3 * n / 4; w[1] = n / 4; xs[1] = x0 + 3 * n / 4; ys[1] = y0; break;
        }
        cu->n_pu = np;
        int try_skip = part == 0 && rnd_n(r, fuzz ? 4 : 3) == 0;
        for (int k = 0; k < np; k++) { Pu *pu = &cu->pu[k]; memset(pu, 0, sizeof *pu); pu->x = xs[k]; pu->y = ys[k]; pu->w = w[k]; pu->h = h[k];
            decide_pu(e, cu, pu, k, try_skip); }
        for (int k = 0; k < np; k++) for (int c = 0; c < 3; c++) { int sc = c ? 1 : 0;
            inter_block(e, &cu->pu[k].m, c, xs[k], ys[k], w[k], h[k], e->cur->pl[c] + (ys[k] >> sc) * e->cur->stride[c] + (xs[k] >> sc), e->cur->stride[c]); }
        if (try_skip && (fuzz ? rnd_n(r, 2) : sad_block(e->cur->pl[0] + y0 * e->cur->stride[0] + x0, e->cur->stride[0],
            e->src.pl[0] + y0 * e->src.stride[0] + x0, e->src.stride[0], n, n) < n * n * 3)) cu->skip = 1;
    }
    if (!cu->pcm && !cu->skip) {
        cu->max_depth = cu->intra ? p->depth_intra + cu->intra_split : p->depth_inter;
        cu->root = decide_tt(e, cu, x0, y0, x0, y0, log2, 0, 0);
        const Tu *rt = &cu->tu[cu->root];
        cu->root_cbf = rt->cbf[0] | rt->cbf[1] | rt->cbf[2];
        if (!cu->intra && !cu->root_cbf && cu->part == 0 && cu->pu[0].merge) cu->skip = 1;
        /* a 2Nx2N merge CU without residual can only be sent as a skipped CU */
    }
    if (cu->skip) for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) { e->skip[I4(e, x, y)] = 1; e->cbf[I4(e, x, y)] = 0; }
    /* QpY of the CU: the delta only exists when it could be sent */
    int sent = p->dqp && !e->dqp_coded && !cu->skip && !cu->pcm && cu->root_cbf;
    if (p->dqp && !e->dqp_coded && !sent) cu->qp = pred;
    /* a CU that was quantised with pred + delta but ends up without coefficients has nothing that depends on the delta */
    /* ---- syntax (7.3.8.5) ---- */
    if (p->bypass) cab_enc(cb, HG_CTX_CU_TQ_BYPASS, cu->bypass);
    if (s->type != 2) {
        int inc = (avail(e, x0, y0, x0 - 1, y0) && e->skip[I4(e, x0 - 1, y0)]) + (avail(e, x0, y0, x0, y0 - 1) && e->skip[I4(e, x0, y0 - 1)]);
        cab_enc(cb, HG_CTX_CU_SKIP + inc, cu->skip);
    }
    if (cu->skip) write_pu(e, cu, &cu->pu[0]);
    else {
        if (s->type != 2) cab_enc(cb, HG_CTX_PRED_MODE, cu->intra);
        if (!cu->intra || log2 == p->min_cb_log2) {
            int pm = cu->part;
            if (cu->intra) cab_enc(cb, HG_CTX_PART_MODE, pm == 0);
            else {
                cab_enc(cb, HG_CTX_PART_MODE, pm == 0);
                if (pm != 0) {
                    if (log2 == p->min_cb_log2) { cab_enc(cb, HG_CTX_PART_MODE + 1, pm == 1);
                        if (pm != 1 && log2 > 3) cab_enc(cb, HG_CTX_PART_MODE + 2, pm == 2); }
                    else if (!p->amp) cab_enc(cb, HG_CTX_PART_MODE + 1, pm == 1);
                    else {
                        int hor = pm == 1 || pm == 4 || pm == 5;
                        cab_enc(cb, HG_CTX_PART_MODE + 1, hor);
                        cab_enc(cb, HG_CTX_PART_MODE + 3, pm == 1 || pm == 2);
                        if (pm >= 4) cab_byp(cb, pm == 5 || pm == 7);
                    }
                }
            }
        }
        if (cu->intra) {
            if (cu->part == 0 && p->pcm && log2 <= MIN(5, p->ctb_log2)) cab_term(cb, cu->pcm);
            if (cu->pcm) {
                BitW *w = cb->w;
                while (w->nbits) bw_put(w, 1, 0);                       /* pcm_alignment_zero_bit */
                for (int c = 0; c < 3; c++) { int sc = c ? 1 : 0, bits = p->pcm == 3 ? 8 : (c ? 6 : 7);
                    for (int y = y0 >> sc; y < (y0 + n) >> sc; y++) for (int x = x0 >> sc; x < (x0 + n) >> sc; x++) bw_put(w, bits,
                    (uint32_t)(e->cur->pl[c][y * e->cur->stride[c] + x] >> (8 - bits))); }
                cab_start(cb, w);
            } else {
                int np = cu->part == 3 ? 4 : 1;
                for (int k = 0; k < np; k++) cab_enc(cb, HG_CTX_PREV_INTRA, cu->prev_flag[k]);
                for (int k = 0; k < np; k++) { if (cu->prev_flag[k]) { cab_byp(cb, cu->mpm_idx[k] > 0);
                    if (cu->mpm_idx[k] > 0) cab_byp(cb, cu->mpm_idx[k] > 1); } else cab_byp_n(cb, 5, (uint32_t)cu->rem[k]); }
                cab_enc(cb, HG_CTX_INTRA_CHROMA, cu->chroma_idx != 4);
                if (cu->chroma_idx != 4) cab_byp_n(cb, 2, (uint32_t)cu->chroma_idx);
            }
        } else for (int k = 0; k < cu->n_pu; k++) write_pu(e, cu, &cu->pu[k]);
        if (!cu->pcm) {
            if (!cu->intra && !(cu->part == 0 && cu->pu[0].merge)) cab_enc(cb, HG_CTX_RQT_ROOT_CBF, cu->root_cbf);
            if (cu->intra || cu->root_cbf) write_tt(e, cu, cu->root, 1, 1);
        }
    }
    if (p->dqp && !e->dqp_coded) cu->qp = pred;                          /* the delta was not sent after all */
    for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) e->qpmap[I4(e, x, y)] = (int8_t)cu->qp;
    e->last_cu_qp = cu->qp; e->qg_open = 1;
    TRACE("CU %d %d %d qp %d\n", x0, y0, log2, cu->qp);
}

/* ------------------------------ coding quadtree, SAO syntax, CTU ------------------------------ */
static void encode_cqt(Enc *e, int x0, int y0, int log2, int depth) {
    const HevcGenParams *p = &e->p; Rng *r = &e->rng;
    int n = 1 << log2, split;
    if (x0 + n <= e->W && y0 + n <= e->H && log2 > p->min_cb_log2) {
        if (p->mode == 1) split = rnd_n(r, 2);
        else {                                                          /* split where the source is busy */
            const uint8_t *sp = e->src.pl[0] + y0 * e->src.stride[0] + x0; int act = 0;
            for (int y = 0; y < n; y += 2) for (int x = 0; x + 2 < n; x += 2) act += ABS(sp[y * e->src.stride[0] + x] - sp[y * e->src.stride[0] + x + 2]);
            split = act * 4 > n * n * (4 + 2 * depth) || rnd_n(r, 8) == 0;
        }
        int inc = (avail(e, x0, y0, x0 - 1, y0) && e->depth[I4(e, x0 - 1, y0)] > depth) + (avail(e, x0, y0, x0, y0 - 1) && e->depth[I4(e, x0, y0 - 1)] > depth);
        cab_enc(&e->cab, HG_CTX_SPLIT_CU + inc, split);
    } else split = log2 > p->min_cb_log2;
    if (p->dqp && log2 >= p->ctb_log2 - (p->dqp - 1)) { e->dqp_coded = 0; e->dqp_val = 0; if (e->qg_open) { e->qp_prev = e->last_cu_qp; e->first_qg = 0; } }
        /* qg_open: a coding unit was coded since the last reset */
    if (split) {
        int h = n >> 1;
        for (int k = 0; k < 4; k++) { int x = x0 + (k & 1) * h, y = y0 + (k >> 1) * h; if (x < e->W && y < e->H) encode_cqt(e, x, y, log2 - 1, depth + 1); }
        return;
    }
    for (int y = y0; y < y0 + n; y += 4) for (int x = x0; x < x0 + n; x += 4) e->depth[I4(e, x, y)] = (uint8_t)depth;
    encode_cu(e, x0, y0, log2);
}
static void encode_sao(Enc *e, int rs) {
    const Slc *s = e->sl; Cab *cb = &e->cab; Rng *r = &e->rng; Sao *o = &e->sao[rs];
    memset(o, 0, sizeof *o);
    if (!s->sao_l && !s->sao_c) return;
    int rx = rs % e->ctb_w, ry = rs / e->ctb_w, left_ok = 0, up_ok = 0, ml = 0, mu = 0;
    if (rx > 0) left_ok = e->ctb_slice[rs - 1] == e->seg_first_ctb && e->tile_of[e->rs2ts[rs - 1]] == e->tile_of[e->rs2ts[rs]];
    if (ry > 0) up_ok = e->ctb_slice[rs - e->ctb_w] == e->seg_first_ctb && e->tile_of[e->rs2ts[rs - e->ctb_w]] == e->tile_of[e->rs2ts[rs]];
    if (left_ok) { ml = rnd_n(r, 3) == 0; cab_enc(cb, HG_CTX_SAO_MERGE, ml); }
    if (up_ok && !ml) { mu = rnd_n(r, 3) == 0; cab_enc(cb, HG_CTX_SAO_MERGE, mu); }
    if (ml) *o = e->sao[rs - 1];
    else if (mu) *o = e->sao[rs - e->ctb_w];
    else for (int c = 0; c < 3; c++) {
        if (!(c ? s->sao_c : s->sao_l)) continue;
        if (c < 2) { o->type[c] = rnd_n(r, 3); cab_enc(cb, HG_CTX_SAO_TYPE, o->type[c] != 0); if (o->type[c]) cab_byp(cb, o->type[c] == 2); }
        else o->type[2] = o->type[1];
        if (!o->type[c]) continue;
        int big = e->p.mode == 1 && rnd_n(r, 4) == 0;
        for (int i = 0; i < 4; i++) { int a = big ? rnd_n(r, 8) : rnd_n(r, 3); for (int k = 0; k < a; k++) cab_byp(cb, 1); if (a < 7) cab_byp(cb, 0);
            o->off[c][i] = a; }
        if (o->type[c] == 1) {
            for (int i = 0; i < 4; i++) if (o->off[c][i]) { int neg = rnd_n(r, 2); cab_byp(cb, neg); if (neg) o->off[c][i] = -o->off[c][i]; }
            o->band[c] = rnd_n(r, 32); cab_byp_n(cb, 5, (uint32_t)o->band[c]);
        } else {
            o->off[c][2] = -o->off[c][2]; o->off[c][3] = -o->off[c][3];
            if (c < 2) { o->eo[c] = rnd_n(r, 4); cab_byp_n(cb, 2, (uint32_t)o->eo[c]); } else o->eo[2] = o->eo[1];
        }
    }
    if (!s->sao_l) o->type[0] = 0;
    if (!s->sao_c) o->type[1] = o->type[2] = 0;
}

/* ------------------------------ in-loop filters (8.7), generator's own statement ------------------------------ */
static int strength(const Enc *e, int xq, int yq, int vertical) {      /* bS of the 4-sample edge piece starting at (xq, yq) */
    int xp = xq - (vertical ? 1 : 0), yp = yq - (vertical ? 0 : 1), q = I4(e, xq, yq), p = I4(e, xp, yp);
    int is_tu = e->edges[q] & (vertical ? 1 : 2), is_pu = e->edges[q] & (vertical ? 4 : 8);
    if (!is_tu && !is_pu) return 0;
    const Slc *sq = &e->slices[e->slice_of[q]], *sp = &e->slices[e->slice_of[p]];
    if (sq->deblock_off) return 0;
    if (sq->addr != sp->addr && !sq->lf_across) return 0;
    { int cl = e->p.ctb_log2, cq = (yq >> cl) * e->ctb_w + (xq >> cl), cp = (yp >> cl) * e->ctb_w + (xp >> cl);
        if (e->tile_of[e->rs2ts[cq]] != e->tile_of[e->rs2ts[cp]] && !e->lf_across_tiles) return 0; }
    if (e->pm[q] == 2 || e->pm[p] == 2) return 2;
    if (is_tu && (e->cbf[q] || e->cbf[p])) return 1;
    const Mot *a = &e->mot[q], *b = &e->mot[p];
    const Pic *ra[2] = {0, 0}, *rb[2] = {0, 0}; const int16_t *va[2] = {0, 0}, *vb[2] = {0, 0}; int na = 0, nb = 0;
    for (int l = 0; l < 2; l++) { if ((a->pf >> l) & 1) { ra[na] = sq->ref[l][a->ref[l]]; va[na++] = a->mv[l]; } if ((b->pf >> l) & 1) {
        rb[nb] = sp->ref[l][b->ref[l]]; vb[nb++] = b->mv[l]; } }
    if (na != nb) return 1;
#define DIFF4(u, v) (ABS((u)[0] - (v)[0]) > 3 || ABS((u)[1] - (v)[1]) > 3)
    if (na == 1) return ra[0] != rb[0] || DIFF4(va[0], vb[0]);
    int straight = ra[0] == rb[0] && ra[1] == rb[1], crossed = ra[0] == rb[1] && ra[1] == rb[0];
    if (!straight && !crossed) return 1;
    int ds = DIFF4(va[0], vb[0]) || DIFF4(va[1], vb[1]), dc = DIFF4(va[0], vb[1]) || DIFF4(va[1], vb[0]);
    if (straight && crossed) return ds && dc;
    return straight ? ds : dc;
#undef DIFF4
}
static void deblock_picture(Enc *e) {
    Pic *pic = e->cur;
    for (int vertical = 1; vertical >= 0; vertical--) {
        int st = pic->stride[0];
        for (int y = 0; y < e->H; y += vertical ? 4 : 8) for (int x = 0; x < e->W; x += vertical ? 8 : 4) {
            if ((vertical ? x : y) == 0) continue;
            int bs = strength(e, x, y, vertical);
            if (!bs) continue;
            int xp = x - (vertical ? 1 : 0), yp = y - (vertical ? 0 : 1);
            const Slc *sq = &e->slices[e->slice_of[I4(e, x, y)]];
            int qp = (e->qpmap[I4(e, x, y)] + e->qpmap[I4(e, xp, yp)] + 1) >> 1;
            int beta = hg_beta_tab[CLIP3(0, 51, qp + 2 * sq->beta)], tc = hg_tc_tab[CLIP3(0, 53, qp + 2 * (bs - 1) + 2 * sq->tc)];
            int across = vertical ? 1 : st, along = vertical ? st : 1;
            uint8_t *q = pic->pl[0] + y * st + x;
            int keep_p = e->nofilt[I4(e, xp, yp)], keep_q = e->nofilt[I4(e, x, y)];
            int d2[2][2];                                               /* second differences of lines 0 and 3, p side / q side */
            for (int k = 0; k < 2; k++) { const uint8_t *l = q + 3 * k * along; d2[k][0] = ABS(l[-3 * across] - 2 * l[-2 * across] + l[-across]);
                d2[k][1] = ABS(l[2 * across] - 2 * l[across] + l[0]); }
            int d0 = d2[0][0] + d2[0][1], d3 = d2[1][0] + d2[1][1];
            if (d0 + d3 >= beta) continue;
            int strong = 1;
            for (int k = 0; k < 2; k++) { const uint8_t *l = q + 3 * k * along; int dk = k ? d3 : d0;
                if (!(2 * dk < (beta >> 2) && ABS(l[-4 * across] - l[-across]) + ABS(l[0] - l[3 * across]) < (beta >> 3) &&
                    ABS(l[-across] - l[0]) < ((5 * tc + 1) >> 1))) strong = 0; }
            int side_thr = (beta + (beta >> 1)) >> 3, mod_p1 = d2[0][0] + d2[1][0] < side_thr, mod_q1 = d2[0][1] + d2[1][1] < side_thr;
            for (int k = 0; k < 4; k++) {
                uint8_t *l = q + k * along;
                int p3 = l[-4 * across], p2 = l[-3 * across], p1 = l[-2 * across], p0 = l[-across], q0 = l[0], q1 = l[across], q2 = l[2 * across],
                    q3 = l[3 * across];
                if (strong) {
                    if (!keep_p) { l[-across] = (uint8_t)CLIP3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
                        l[-2 * across] = (uint8_t)CLIP3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
                        l[-3 * across] = (uint8_t)CLIP3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3); }
                    if (!keep_q) { l[0] = (uint8_t)CLIP3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
                        l[across] = (uint8_t)CLIP3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
                        l[2 * across] = (uint8_t)CLIP3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3); }
                } else {
                    int dl = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
                    if (ABS(dl) >= 10 * tc) continue;
                    dl = CLIP3(-tc, tc, dl);
                    if (!keep_p) { l[-across] = (uint8_t)CLIP1(p0 + dl); if (mod_p1) l[-2 * across] = (uint8_t)CLIP1(p1 + CLIP3(-(tc >> 1), tc >> 1,
                        (((p2 + p0 + 1) >> 1) - p1 + dl) >> 1)); }
                    if (!keep_q) { l[0] = (uint8_t)CLIP1(q0 - dl); if (mod_q1) l[across] = (uint8_t)CLIP1(q1 + CLIP3(-(tc >> 1), tc >> 1,
                        (((q2 + q0 + 1) >> 1) - q1 - dl) >> 1)); }
                }
            }
        }
        for (int c = 1; c < 3; c++) {
            int stc = pic->stride[c], qoff = c == 1 ? e->p.cb_qp_off : e->p.cr_qp_off;
            for (int y = 0; y < e->H / 2; y += vertical ? 4 : 8) for (int x = 0; x < e->W / 2; x += vertical ? 8 : 4) {
                if ((vertical ? x : y) == 0) continue;
                int xl = 2 * x, yl = 2 * y;
                if (strength(e, xl, yl, vertical) != 2) continue;
                int xp = xl - (vertical ? 1 : 0), yp = yl - (vertical ? 0 : 1);
                const Slc *sq = &e->slices[e->slice_of[I4(e, xl, yl)]];
                int qpc = hg_qpc_tab[CLIP3(0, 57, ((e->qpmap[I4(e, xl, yl)] + e->qpmap[I4(e, xp, yp)] + 1) >> 1) + qoff)];
                int tc = hg_tc_tab[CLIP3(0, 53, qpc + 2 + 2 * sq->tc)], across = vertical ? 1 : stc, along = vertical ? stc : 1;
                int keep_p = e->nofilt[I4(e, xp, yp)], keep_q = e->nofilt[I4(e, xl, yl)];
                for (int k = 0; k < 4; k++) {
                    uint8_t *l = pic->pl[c] + y * stc + x + k * along;
                    int p1 = l[-2 * across], p0 = l[-across], q0 = l[0], q1 = l[across];
                    int dl = CLIP3(-tc, tc, (((q0 - p0) << 2) + p1 - q1 + 4) >> 3);
                    if (!keep_p) l[-across] = (uint8_t)CLIP1(p0 + dl);
                    if (!keep_q) l[0] = (uint8_t)CLIP1(q0 - dl);
                }
            }
        }
    }
}
static void sao_picture(Enc *e) {
    Pic *pic = e->cur;
    if (!e->p.sao) return;
    for (int c = 0; c < 3; c++) memcpy(e->dbk[c], pic->pl[c], (size_t)pic->stride[c] * (size_t)(e->H >> (c ? 1 : 0)));
    for (int c = 0; c < 3; c++) {
        int sc = c ? 1 : 0, pw = e->W >> sc, ph = e->H >> sc, cs = e->ctb >> sc, st = pic->stride[c];
        for (int rs = 0; rs < e->ctb_w * e->ctb_h; rs++) {
            const Sao *o = &e->sao[rs];
            if (!o->type[c]) continue;
            int x0 = (rs % e->ctb_w) * cs, y0 = (rs / e->ctb_w) * cs;
            for (int y = y0; y < MIN(y0 + cs, ph); y++) for (int x = x0; x < MIN(x0 + cs, pw); x++) {
                if (e->nofilt[I4(e, x << sc, y << sc)]) continue;
                int v = e->dbk[c][y * st + x], add = 0;
                if (o->type[c] == 1) { int k = ((v >> 3) - o->band[c]) & 31; if (k < 4) add = o->off[c][k]; }
                else {
                    static const int dx[4] = {1, 0, 1, -1}, dy[4] = {0, 1, 1, 1};    /* second neighbour; the first is its mirror image */
                    int xa = x - dx[o->eo[c]], ya = y - dy[o->eo[c]], xb = x + dx[o->eo[c]], yb = y + dy[o->eo[c]];
                    if (xa < 0 || xb < 0 || xa >= pw || xb >= pw || ya < 0 || yb >= ph) continue;
                    int bad = 0;
                    for (int k = 0; k < 2; k++) {
                        int xn = (k ? xb : xa) << sc, yn = (k ? yb : ya) << sc, xc = x << sc, yc = y << sc;
                        const Slc *sn = &e->slices[e->slice_of[I4(e, xn, yn)]], *sc_ = &e->slices[e->slice_of[I4(e, xc, yc)]];
                        if (sn->addr != sc_->addr) { int earlier = zrank(e, xn, yn) < zrank(e, xc, yc);
                            if (earlier ? !sc_->lf_across : !sn->lf_across) bad = 1; }
                        int cl = e->p.ctb_log2;
                        if (e->tile_of[e->rs2ts[(yn >> cl) * e->ctb_w + (xn >> cl)]] != e->tile_of[e->rs2ts[(yc >> cl) * e->ctb_w + (xc >> cl)]] &&
                            !e->lf_across_tiles) bad = 1;
                    }
                    if (bad) continue;
                    int a = e->dbk[c][ya * st + xa], b = e->dbk[c][yb * st + xb], sgn = SIGN(v - a) + SIGN(v - b);
                    add = sgn == -2 ? o->off[c][0] : sgn == -1 ? o->off[c][1] : sgn == 1 ? o->off[c][2] : sgn == 2 ? o->off[c][3] : 0;
                }
                pic->pl[c][y * st + x] = (uint8_t)CLIP1(v + add);
            }
        }
    }
}

/* ------------------------------ scaling lists (7.3.4 / 7.4.5) ------------------------------ */
static void diag_order(int n, int *xs, int *ys) { int k = 0; for (int s = 0; s < 2 * n - 1; s++) for (int x = 0; x <= s; x++) { int y = s - x;
    if (x < n && y < n) { xs[k] = x; ys[k] = y; k++; } } }
static void lists_default(Enc *e) {
    memset(e->sl4, 16, sizeof e->sl4);
    for (int m = 0; m < 6; m++) { memcpy(e->sl8[m], hg_scaling_default[m >= 3], 64); memcpy(e->sl16[m], hg_scaling_default[m >= 3], 64);
        memcpy(e->sl32[m], hg_scaling_default[m >= 3], 64); e->dc16[m] = e->dc32[m] = 16; }
}
static void lists_expand(Enc *e) {
    int x4[16], y4[16], x8[64], y8[64]; diag_order(4, x4, y4); diag_order(8, x8, y8);
    for (int m = 0; m < 6; m++) {
        for (int i = 0; i < 16; i++) e->sf[0][m][y4[i] * 4 + x4[i]] = e->sl4[m][i];
        for (int i = 0; i < 64; i++) {
            e->sf[1][m][y8[i] * 8 + x8[i]] = e->sl8[m][i];
            for (int j = 0; j < 4; j++) e->sf[2][m][(y8[i] * 2 + (j >> 1)) * 16 + x8[i] * 2 + (j & 1)] = e->sl16[m][i];
            if (m < 2) for (int j = 0; j < 16; j++) e->sf[3][m][(y8[i] * 4 + (j >> 2)) * 32 + x8[i] * 4 + (j & 3)] = e->sl32[m * 3][i];
        }
        e->sf[2][m][0] = e->dc16[m]; if (m < 2) e->sf[3][m][0] = e->dc32[m * 3];
    }
}
/* random lists + their syntax */
static void write_scaling_list_data(Enc *e, BitW *w) {
    Rng *r = &e->rng;
    lists_default(e);
    for (int size = 0; size < 4; size++) for (int k = 0; k < (size == 3 ? 2 : 6); k++) {
        int m = size == 3 ? 3 * k : k, n = size ? 64 : 16;
        uint8_t *list = size == 0 ? e->sl4[m] : size == 1 ? e->sl8[m] : size == 2 ? e->sl16[m] : e->sl32[m];
        uint8_t *dc = size == 2 ? &e->dc16[m] : size == 3 ? &e->dc32[m] : NULL;
        int choice = rnd_n(r, 3);
        if (choice == 0) {                                              /* copy from an earlier matrix of this size, or the default (delta 0) */
            int delta = rnd_n(r, k + 1);
            bw_put(w, 1, 0); bw_ue(w, (uint32_t)delta);
            if (delta == 0) { if (size == 0) memset(list, 16, 16); else memcpy(list, hg_scaling_default[size == 3 ? k : (m >= 3)], 64); if (dc) *dc = 16; }
            else { int rm = size == 3 ? 3 * (k - delta) : m - delta;
                const uint8_t *src = size == 0 ? e->sl4[rm] : size == 1 ? e->sl8[rm] : size == 2 ? e->sl16[rm] : e->sl32[rm]; memmove(list, src, (size_t)n);
                if (dc) *dc = size == 2 ? e->dc16[rm] : e->dc32[rm]; }
        } else {
            bw_put(w, 1, 1);
            int next = 8;
            if (size > 1) { int d = 4 + rnd_n(r, 60); bw_se(w, d - 8); next = d; *dc = (uint8_t)d; }
            for (int i = 0; i < n; i++) { int v = choice == 1 ? 8 + i / 2 + rnd_n(r, 9) : 1 + rnd_n(r, 255); int dl = v - next; if (dl > 127) dl -= 256;
                if (dl < -128) dl += 256; bw_se(w, dl); next = v; list[i] = (uint8_t)v; }
        }
    }
}

/* ------------------------------ coded video sequence plan ------------------------------ */
typedef struct { int t, poc, type /* 0 B 1 P 2 I */, is_ref, idr, n[2], l[2][4], lt, cra, rasl; } Sched;

static void plan_sequence(const HevcGenParams *p, Sched *sc, int *count) {
    int n = 0, last_idr_t = 0, carry_prev = -1;
    for (int t0 = 0; t0 < p->frames; t0 += p->intra_period) {
        int end = MIN(p->frames, t0 + p->intra_period);
        int g = p->gop == 8 ? 8 : p->gop + 1;
        int open = p->open_gop && t0 > 0 && p->gop >= 1 && p->gop <= 3 && carry_prev >= 0;
        Sched *s = &sc[n++]; memset(s, 0, sizeof *s); s->t = t0; s->type = 2; s->is_ref = 1; s->idr = !open; s->cra = open;
        if (!open) last_idr_t = t0;
        if (open) for (int t = carry_prev + 1; t < t0; t++) {          /* leading pictures: displayed before the CRA, decoded after it */
            Sched *b = &sc[n++]; memset(b, 0, sizeof *b); b->t = t; b->type = 0; b->is_ref = 0; b->rasl = 1;
            b->l[0][b->n[0]++] = carry_prev; if (p->num_ref > 1) b->l[0][b->n[0]++] = t0;
            b->l[1][b->n[1]++] = t0; if (p->num_ref > 1) b->l[1][b->n[1]++] = carry_prev;
        }
        carry_prev = -1;
        int last_anchor = t0;
        int anchors[4] = { t0, -1, -1, -1 };
        for (int a = t0 + g; ; a += g) {
            if (a >= end) {
                /* the next period opens with a CRA at `end`: the pictures between the last anchor and it wait for that CRA */
                if (p->open_gop && p->gop >= 1 && p->gop <= 3 && end < p->frames && end - last_anchor == g) { carry_prev = last_anchor; break; }
                for (int t = last_anchor + 1; t < end; t++) {           /* tail: plain P pictures in display order */
                    s = &sc[n++]; memset(s, 0, sizeof *s); s->t = t; s->type = 1; s->is_ref = 1;
                    for (int k = 0; k < p->num_ref && k < 4; k++) if (anchors[k] >= 0) s->l[0][s->n[0]++] = anchors[k];
                    for (int k = 3; k > 0; k--) anchors[k] = anchors[k - 1];
                    anchors[0] = t;
                }
                break;
            }
            s = &sc[n++]; memset(s, 0, sizeof *s); s->t = a; s->type = 1; s->is_ref = 1;
            for (int k = 0; k < p->num_ref && k < 4; k++) if (anchors[k] >= 0) s->l[0][s->n[0]++] = anchors[k];
            if (p->lt_ref && p->gop == 0 && a >= t0 + 3 && s->n[0] < 4) { int dup = 0; for (int k = 0; k < s->n[0]; k++) dup |= s->l[0][k] == t0; if (!dup) {
                s->l[0][s->n[0]++] = t0; s->lt = 1; } }
            int prev = anchors[0];
            for (int k = 3; k > 0; k--) anchors[k] = anchors[k - 1];
            anchors[0] = a;
            last_anchor = a;
            if (p->gop >= 1 && p->gop <= 3) for (int t = prev + 1; t < a; t++) {
                s = &sc[n++]; memset(s, 0, sizeof *s); s->t = t; s->type = 0; s->is_ref = 0;
                s->l[0][s->n[0]++] = prev; if (p->num_ref > 1) s->l[0][s->n[0]++] = a;
                s->l[1][s->n[1]++] = a; if (p->num_ref > 1) s->l[1][s->n[1]++] = prev;
            } else if (p->gop == 8) {
                static const int off[7] = {4, 2, 1, 3, 6, 5, 7}, isref[7] = {1, 1, 0, 0, 1, 0, 0};
                static const int l0[7][2] = {{0, -1}, {0, -1}, {0, -1}, {2, 0}, {4, 0}, {4, 0}, {6, 4}}, l1[7][2] = {{8, -1}, {4, 8}, {2, 4}, {4, 8}, {8, -1},
                    {6, 8}, {8, -1}};
                for (int k = 0; k < 7; k++) {
                    s = &sc[n++]; memset(s, 0, sizeof *s); s->t = prev + off[k]; s->type = 0; s->is_ref = isref[k];
                    for (int j = 0; j < 2; j++) { if (l0[k][j] >= 0 && (j == 0 || p->num_ref > 1)) s->l[0][s->n[0]++] = prev + l0[k][j];
                        if (l1[k][j] >= 0 && (j == 0 || p->num_ref > 1)) s->l[1][s->n[1]++] = prev + l1[k][j]; }
                }
            }
        }
        (void)last_idr_t;
    }
    /* POC: display distance from the last IDR picture (decoding order: an IDR precedes everything of its period) */
    for (int i = 0, idr_t = 0; i < n; i++) { if (sc[i].idr) idr_t = sc[i].t; sc[i].poc = sc[i].t - idr_t; }
    *count = n;
}

/* short-term RPS of picture idx: every earlier reference picture (since the last IDR) that this or a later picture still lists */
static void plan_rps(const Sched *sched, int n_sched, int idx, RpsSet *o) {
    memset(o, 0, sizeof *o);
    const Sched *sc = &sched[idx];
    int period0 = idx; while (!sched[period0].idr) period0--;
    for (int pass = 0; pass < 2; pass++) for (int dist = 1; dist < 64; dist++) for (int j = period0; j < idx; j++) {
        if (!sched[j].is_ref || (sc->lt && sched[j].idr)) continue;
        int d = sched[j].poc - sc->poc;
        if ((pass == 0 ? -d : d) != dist) continue;
        int needed = 0, used = 0;
        for (int k = idx; k < n_sched &&
            !sched[k].idr; k++) for (int l = 0; l < 2; l++) for (int i = 0; i < sched[k].n[l];
            i++) if (sched[k].l[l][i] == sched[j].t) { needed = 1; if (k == idx) used = 1; }
        if (!needed) continue;
        if (pass == 0) { o->dneg[o->n_neg] = d; o->uneg[o->n_neg++] = used; } else { o->dpos[o->n_pos] = d; o->upos[o->n_pos++] = used; }
    }
}
static int rps_equal(const RpsSet *a, const RpsSet *b) {
    if (a->n_neg != b->n_neg || a->n_pos != b->n_pos) return 0;
    for (int i = 0; i < a->n_neg; i++) if (a->dneg[i] != b->dneg[i] || a->uneg[i] != b->uneg[i]) return 0;
    for (int i = 0; i < a->n_pos; i++) if (a->dpos[i] != b->dpos[i] || a->upos[i] != b->upos[i]) return 0;
    return 1;
}
static void rps_write_explicit(BitW *w, const RpsSet *t) {
    bw_ue(w, (uint32_t)t->n_neg); bw_ue(w, (uint32_t)t->n_pos);
    for (int i = 0, prev = 0; i < t->n_neg; i++) { bw_ue(w, (uint32_t)(prev - t->dneg[i] - 1)); bw_put(w, 1, (uint32_t)t->uneg[i]); prev = t->dneg[i]; }
    for (int i = 0, prev = 0; i < t->n_pos; i++) { bw_ue(w, (uint32_t)(t->dpos[i] - prev - 1)); bw_put(w, 1, (uint32_t)t->upos[i]); prev = t->dpos[i]; }
}
/* inter RPS prediction (7.3.7 / 7.4.8) of set t from set r: returns 1 and writes delta_rps + flags when t can be expressed that way */
static int rps_write_inter(BitW *w, const RpsSet *t, const RpsSet *r) {
    int nr = r->n_neg + r->n_pos, rd[33];
    for (int j = 0; j < r->n_neg; j++) rd[j] = r->dneg[j];
    for (int j = 0; j < r->n_pos; j++) rd[r->n_neg + j] = r->dpos[j];
    rd[nr] = 0;
    int nt = t->n_neg + t->n_pos, td[32], tu[32];
    for (int j = 0; j < t->n_neg; j++) { td[j] = t->dneg[j]; tu[j] = t->uneg[j]; }
    for (int j = 0; j < t->n_pos; j++) { td[t->n_neg + j] = t->dpos[j]; tu[t->n_neg + j] = t->upos[j]; }
    for (int a = 0; a < nt; a++) for (int b = 0; b <= nr; b++) {
        int d = td[a] - rd[b];
        if (d == 0 || d < -32768 || d > 32767) continue;
        int used[33], keep[33], hit = 0;
        for (int j = 0; j <= nr; j++) { used[j] = keep[j] = 0; for (int k = 0; k < nt; k++) if (td[k] == rd[j] + d) { keep[j] = 1; used[j] = tu[k]; hit++; } }
        if (hit != nt) continue;
        /* derive exactly as a decoder would (the order of the result matters) and compare */
        RpsSet o; memset(&o, 0, sizeof o);
        int i = 0;
        for (int j = r->n_pos - 1; j >= 0; j--) { int dp = r->dpos[j] + d; if (dp < 0 && keep[r->n_neg + j]) { o.dneg[i] = dp;
            o.uneg[i++] = used[r->n_neg + j]; } }
        if (d < 0 && keep[nr]) { o.dneg[i] = d; o.uneg[i++] = used[nr]; }
        for (int j = 0; j < r->n_neg; j++) { int dp = r->dneg[j] + d; if (dp < 0 && keep[j]) { o.dneg[i] = dp; o.uneg[i++] = used[j]; } }
        o.n_neg = i; i = 0;
        for (int j = r->n_neg - 1; j >= 0; j--) { int dp = r->dneg[j] + d; if (dp > 0 && keep[j]) { o.dpos[i] = dp; o.upos[i++] = used[j]; } }
        if (d > 0 && keep[nr]) { o.dpos[i] = d; o.upos[i++] = used[nr]; }
        for (int j = 0; j < r->n_pos; j++) { int dp = r->dpos[j] + d; if (dp > 0 && keep[r->n_neg + j]) { o.dpos[i] = dp; o.upos[i++] = used[r->n_neg + j]; } }
        o.n_pos = i;
        if (!rps_equal(&o, t)) continue;
        bw_put(w, 1, d < 0); bw_ue(w, (uint32_t)(ABS(d) - 1));
        for (int j = 0; j <= nr; j++) { bw_put(w, 1, (uint32_t)(keep[j] && used[j])); if (!(keep[j] && used[j])) bw_put(w, 1, (uint32_t)keep[j]); }
        return 1;
    }
    return 0;
}

/* ------------------------------ parameter sets (7.3.2) ------------------------------ */
static void write_ptl(BitW *w) {
    bw_put(w, 2, 0); bw_put(w, 1, 0); bw_put(w, 5, 1);               /* Main profile, main tier */
    bw_put(w, 32, 0x60000000u);                                        /* compatibility flags: Main (1) and Main 10 (2) */
    bw_put(w, 1, 1); bw_put(w, 1, 0); bw_put(w, 1, 0); bw_put(w, 1, 1);   /* progressive, !interlaced, !non-packed, frame-only */
    bw_put(w, 32, 0); bw_put(w, 12, 0);                               /* 44 reserved bits */
    bw_put(w, 8, 186);                                                 /* level 6.2 */
}
static void write_vps(Enc *e, BitW *out, int max_dpb, int reorder) {
    BitW w; memset(&w, 0, sizeof w);
    bw_put(&w, 4, 0); bw_put(&w, 2, 3); bw_put(&w, 6, 0); bw_put(&w, 3, 0); bw_put(&w, 1, 1); bw_put(&w, 16, 0xFFFF);
    write_ptl(&w);
    bw_put(&w, 1, 1); bw_ue(&w, (uint32_t)(max_dpb - 1)); bw_ue(&w, (uint32_t)reorder); bw_ue(&w, 0);
    bw_put(&w, 6, 0); bw_ue(&w, 0); bw_put(&w, 1, 0); bw_put(&w, 1, 0);
    bw_trailing(&w);
    write_nal(out, 32, 0, w.buf, w.len, NULL, 0); free(w.buf);
}
static void rps_write_explicit(BitW *w, const RpsSet *t);
static int rps_write_inter(BitW *w, const RpsSet *t, const RpsSet *r);
static void write_sps(Enc *e, BitW *out, int max_dpb, int reorder) {
    const HevcGenParams *p = &e->p; BitW w; memset(&w, 0, sizeof w);
    bw_put(&w, 4, 0); bw_put(&w, 3, 0); bw_put(&w, 1, 1);
    write_ptl(&w);
    bw_ue(&w, 0); bw_ue(&w, 1);
    bw_ue(&w, (uint32_t)e->W); bw_ue(&w, (uint32_t)e->H);
    int crop = e->W != p->width || e->H != p->height;
    bw_put(&w, 1, (uint32_t)crop);
    if (crop) { bw_ue(&w, 0); bw_ue(&w, (uint32_t)((e->W - p->width) / 2)); bw_ue(&w, 0); bw_ue(&w, (uint32_t)((e->H - p->height) / 2)); }
    bw_ue(&w, 0); bw_ue(&w, 0);
    bw_ue(&w, (uint32_t)(e->poc_bits - 4));
    bw_put(&w, 1, 1); bw_ue(&w, (uint32_t)(max_dpb - 1)); bw_ue(&w, (uint32_t)reorder); bw_ue(&w, 0);
    bw_ue(&w, (uint32_t)(p->min_cb_log2 - 3)); bw_ue(&w, (uint32_t)(p->ctb_log2 - p->min_cb_log2));
    bw_ue(&w, (uint32_t)(p->min_tb_log2 - 2)); bw_ue(&w, (uint32_t)(p->max_tb_log2 - p->min_tb_log2));
    bw_ue(&w, (uint32_t)p->depth_inter); bw_ue(&w, (uint32_t)p->depth_intra);
    bw_put(&w, 1, p->scaling != 0);
    if (p->scaling) { bw_put(&w, 1, p->scaling == 2); if (p->scaling == 2) write_scaling_list_data(e, &w); }
    bw_put(&w, 1, (uint32_t)p->amp); bw_put(&w, 1, (uint32_t)p->sao); bw_put(&w, 1, p->pcm != 0);
    if (p->pcm) { bw_put(&w, 4, p->pcm == 3 ? 7 : 6); bw_put(&w, 4, p->pcm == 3 ? 7 : 5); bw_ue(&w, (uint32_t)(p->min_cb_log2 - 3));
        bw_ue(&w, (uint32_t)(MIN(5, p->ctb_log2) - p->min_cb_log2)); bw_put(&w, 1, p->pcm == 1 || p->pcm == 3); }
    bw_ue(&w, (uint32_t)e->n_sps_sets);                                /* short-term reference picture sets (none: every slice header carries its own) */
    for (int i = 0; i < e->n_sps_sets; i++) {
        if (i == 0) { rps_write_explicit(&w, &e->sps_sets[0]); continue; }
        BitW t; memset(&t, 0, sizeof t);                                /* try inter prediction from the previous set (7.3.7) */
        size_t len0 = w.len; int nb0 = w.nbits; uint32_t cur0 = w.cur;
        bw_put(&w, 1, 1);
        if (!rps_write_inter(&w, &e->sps_sets[i], &e->sps_sets[i - 1])) { w.len = len0; w.nbits = nb0; w.cur = cur0; bw_put(&w, 1, 0);
            rps_write_explicit(&w, &e->sps_sets[i]); }
        (void)t;
    }
    bw_put(&w, 1, (uint32_t)p->lt_ref);
    if (p->lt_ref) bw_ue(&w, 0);
    bw_put(&w, 1, (uint32_t)p->tmvp); bw_put(&w, 1, (uint32_t)p->strong_intra);
    if (p->vui_fps > 0) {                                             /* vui_parameters() (E.2.1) */
        bw_put(&w, 1, 1);
        bw_put(&w, 1, 1); bw_put(&w, 8, 1);                           /* aspect_ratio_info_present_flag, aspect_ratio_idc 1 */
        bw_put(&w, 1, 0); bw_put(&w, 1, 0); bw_put(&w, 1, 0);         /* overscan, video signal type, chroma location: absent */
        bw_put(&w, 1, 0); bw_put(&w, 1, 0); bw_put(&w, 1, 0);         /* neutral_chroma_indication, field_seq, frame_field_info_present */
        bw_put(&w, 1, 0);                                             /* default_display_window_flag */
        bw_put(&w, 1, 1); bw_put(&w, 32, 1); bw_put(&w, 32, (uint32_t)p->vui_fps);     /* vui_timing_info: num_units_in_tick, time_scale */
        bw_put(&w, 1, 0); bw_put(&w, 1, 0);                           /* poc_proportional_to_timing, vui_hrd_parameters_present */
        bw_put(&w, 1, 0);                                             /* bitstream_restriction_flag */
    } else bw_put(&w, 1, 0);                                          /* no VUI */
    bw_put(&w, 1, 0);                                                 /* no extension */
    bw_trailing(&w);
    write_nal(out, 33, 0, w.buf, w.len, NULL, 0); free(w.buf);
}
static void write_pps(Enc *e, BitW *out) {
    const HevcGenParams *p = &e->p; BitW w; memset(&w, 0, sizeof w);
    bw_ue(&w, 0); bw_ue(&w, 0);
    bw_put(&w, 1, (uint32_t)p->dep_slices); bw_put(&w, 1, 0); bw_put(&w, 3, 0);
    bw_put(&w, 1, (uint32_t)p->sdh); bw_put(&w, 1, (uint32_t)(p->cabac_init != 0));
    bw_ue(&w, (uint32_t)(p->num_ref - 1)); bw_ue(&w, (uint32_t)(p->num_ref - 1));
    bw_se(&w, p->qp - 26 - 2);                                         /* init_qp: slices add slice_qp_delta = +2 */
    bw_put(&w, 1, (uint32_t)p->cip); bw_put(&w, 1, (uint32_t)p->tskip);
    bw_put(&w, 1, p->dqp != 0); if (p->dqp) bw_ue(&w, (uint32_t)(p->dqp - 1));
    bw_se(&w, p->cb_qp_off); bw_se(&w, p->cr_qp_off); bw_put(&w, 1, 0);
    bw_put(&w, 1, p->wp != 0); bw_put(&w, 1, p->wp != 0);
    bw_put(&w, 1, (uint32_t)p->bypass);
    int tiles = p->tile_cols * p->tile_rows > 1;
    bw_put(&w, 1, (uint32_t)tiles); bw_put(&w, 1, (uint32_t)p->wpp);
    if (tiles) {
        bw_ue(&w, (uint32_t)(p->tile_cols - 1)); bw_ue(&w, (uint32_t)(p->tile_rows - 1)); bw_put(&w, 1, !e->tile_explicit);
        if (e->tile_explicit) { for (int i = 0; i + 1 < p->tile_cols; i++) bw_ue(&w, (uint32_t)(e->tile_cb[i + 1] - e->tile_cb[i] - 1));
            for (int i = 0; i + 1 < p->tile_rows; i++) bw_ue(&w, (uint32_t)(e->tile_rb[i + 1] - e->tile_rb[i] - 1)); }
        bw_put(&w, 1, (uint32_t)e->lf_across_tiles);
    }
    bw_put(&w, 1, 1);                                                  /* pps_loop_filter_across_slices_enabled_flag (slices decide) */
    bw_put(&w, 1, 1);                                                  /* deblocking_filter_control_present_flag */
    bw_put(&w, 1, p->deblock == 2); bw_put(&w, 1, p->deblock == 0);
    if (p->deblock != 0) { bw_se(&w, p->deblock == 2 ? 1 : 0); bw_se(&w, p->deblock == 2 ? -1 : 0); }
    bw_put(&w, 1, p->scaling == 3);
    if (p->scaling == 3) write_scaling_list_data(e, &w);
    bw_put(&w, 1, (uint32_t)p->rplm);
    bw_ue(&w, (uint32_t)(p->par_mrg - 2));
    bw_put(&w, 1, 0); bw_put(&w, 1, 0);
    bw_trailing(&w);
    write_nal(out, 34, 0, w.buf, w.len, NULL, 0); free(w.buf);
}

/* ------------------------------ picture coding ------------------------------ */
typedef struct {                 /* picture-level decisions shared by its slices */
    int type, nal, poc, idr;
    int n_neg, n_pos, dneg[16], dpos[16], uneg[16], upos[16];          /* short-term RPS: deltas (closest first) and used_by_curr */
    Pic *before[16], *after[16], *ltc[4]; int nb, na, nl, lt_poc_lsb[4], lt_msb[4], lt_cycle[4];
    int n_total;
} PicPlan;

static Pic *find_poc(Enc *e, int poc) { for (int i = 0; i < 10; i++) if (e->dpb[i].used && e->dpb[i].is_ref && e->dpb[i].poc == poc) return &e->dpb[i];
    return NULL; }

static void write_slice_header(Enc *e, BitW *w, const PicPlan *pp, Slc *s, int first, int dependent, int seg_addr, const size_t *sizes, int n_entry) {
    const HevcGenParams *p = &e->p;
    if (getenv("HG_DBG")) fprintf(stderr,
        "SH type %d poc %d nref %d %d tmvp %d col %d %d mvdl1z %d cabac_init %d wp %d denom %d %d rpsneg %d pos %d nl %d ntotal %d\n", s->type, pp->poc,
        s->n_ref[0], s->n_ref[1], s->tmvp, s->col_l0, s->col_idx, s->mvd_l1_zero, s->cabac_init, s->wp_on, s->wp_denom[0], s->wp_denom[1], pp->n_neg,
        pp->n_pos, pp->nl, pp->n_total);
    bw_put(w, 1, (uint32_t)first);
    if (pp->nal >= 16 && pp->nal <= 23) bw_put(w, 1, 0);
    bw_ue(w, 0);
    if (!first) { if (p->dep_slices) bw_put(w, 1, (uint32_t)dependent); bw_put(w, ceil_log2(e->ctb_w * e->ctb_h), (uint32_t)seg_addr); }
    if (!dependent) {
        bw_ue(w, (uint32_t)s->type);
        if (!pp->idr) {
            bw_put(w, e->poc_bits, (uint32_t)(pp->poc & ((1 << e->poc_bits) - 1)));
            {
                RpsSet t; memset(&t, 0, sizeof t);
                t.n_neg = pp->n_neg; t.n_pos = pp->n_pos;
                for (int i = 0; i < t.n_neg; i++) { t.dneg[i] = pp->dneg[i]; t.uneg[i] = pp->uneg[i]; }
                for (int i = 0; i < t.n_pos; i++) { t.dpos[i] = pp->dpos[i]; t.upos[i] = pp->upos[i]; }
                int match = -1;
                for (int i = 0; i < e->n_sps_sets; i++) if (rps_equal(&e->sps_sets[i], &t)) match = i;
                if (match >= 0 && rnd_n(&e->rng, 4) != 0) {             /* a set of the SPS, by index */
                    bw_put(w, 1, 1);
                    if (e->n_sps_sets > 1) bw_put(w, ceil_log2(e->n_sps_sets), (uint32_t)match);
                } else {
                    bw_put(w, 1, 0);                                    /* st_ref_pic_set(num_short_term_ref_pic_sets) in the slice header */
                    int done = 0;
                    if (e->n_sps_sets > 0) {                            /* its index is not 0, so the inter prediction flag is present */
                        int start = rnd_n(&e->rng, e->n_sps_sets);
                        for (int k = 0; k < e->n_sps_sets && !done; k++) {
                            int ref = (start + k) % e->n_sps_sets;
                            size_t len0 = w->len; int nb0 = w->nbits; uint32_t cur0 = w->cur;
                            bw_put(w, 1, 1); bw_ue(w, (uint32_t)(e->n_sps_sets - 1 - ref));      /* delta_idx_minus1 */
                            if (rps_write_inter(w, &t, &e->sps_sets[ref])) done = 1; else { w->len = len0; w->nbits = nb0; w->cur = cur0; }
                        }
                        if (!done) bw_put(w, 1, 0);
                    }
                    if (!done) rps_write_explicit(w, &t);
                }
            }
            if (p->lt_ref) {
                bw_ue(w, (uint32_t)pp->nl);
                for (int i = 0; i < pp->nl; i++) { bw_put(w, e->poc_bits, (uint32_t)pp->lt_poc_lsb[i]); bw_put(w, 1, 1); bw_put(w, 1, (uint32_t)pp->lt_msb[i]);
                    if (pp->lt_msb[i]) bw_ue(w, (uint32_t)pp->lt_cycle[i]); }
            }
            if (p->tmvp) bw_put(w, 1, (uint32_t)s->tmvp);
        }
        if (p->sao) { bw_put(w, 1, (uint32_t)s->sao_l); bw_put(w, 1, (uint32_t)s->sao_c); }
        if (s->type != 2) {
            int ovr = s->n_ref[0] != p->num_ref || (s->type == 0 && s->n_ref[1] != p->num_ref);
            bw_put(w, 1, (uint32_t)ovr);
            if (ovr) { bw_ue(w, (uint32_t)(s->n_ref[0] - 1)); if (s->type == 0) bw_ue(w, (uint32_t)(s->n_ref[1] - 1)); }
            if (p->rplm && pp->n_total > 1) {
                int nbits = ceil_log2(pp->n_total);
                for (int l = 0; l < (s->type == 0 ? 2 : 1); l++) {
                    bw_put(w, 1, (uint32_t)s->rplm_flag[l]);
                    if (s->rplm_flag[l]) for (int i = 0; i < s->n_ref[l]; i++) bw_put(w, nbits, (uint32_t)s->list_entry[l][i]);
                }
            }
            if (s->type == 0) bw_put(w, 1, (uint32_t)s->mvd_l1_zero);
            if (p->cabac_init) bw_put(w, 1, (uint32_t)s->cabac_init);
            if (s->tmvp) { if (s->type == 0) bw_put(w, 1, (uint32_t)s->col_l0); if ((s->col_l0 ? s->n_ref[0] : s->n_ref[1]) > 1) bw_ue(w,
                (uint32_t)s->col_idx); }
            if (p->wp) {
                bw_ue(w, (uint32_t)s->wp_denom[0]); bw_se(w, s->wp_denom[1] - s->wp_denom[0]);
                for (int l = 0; l < (s->type == 0 ? 2 : 1); l++) {
                    for (int i = 0; i < s->n_ref[l]; i++) bw_put(w, 1, s->wp_w[l][i][0] != (1 << s->wp_denom[0]) || s->wp_o[l][i][0] != 0);
                    for (int i = 0; i < s->n_ref[l]; i++) bw_put(w, 1, s->wp_w[l][i][1] != (1 << s->wp_denom[1]) || s->wp_o[l][i][1] != 0 ||
                        s->wp_w[l][i][2] != (1 << s->wp_denom[1]) || s->wp_o[l][i][2] != 0);
                    for (int i = 0; i < s->n_ref[l]; i++) {
                        if (s->wp_w[l][i][0] != (1 << s->wp_denom[0]) || s->wp_o[l][i][0] != 0) { bw_se(w, s->wp_w[l][i][0] - (1 << s->wp_denom[0]));
                            bw_se(w, s->wp_o[l][i][0]); }
                        if (s->wp_w[l][i][1] != (1 << s->wp_denom[1]) || s->wp_o[l][i][1] != 0 || s->wp_w[l][i][2] != (1 << s->wp_denom[1]) ||
                            s->wp_o[l][i][2] != 0)
                            for (int c = 1; c < 3; c++) { int wgt = s->wp_w[l][i][c]; bw_se(w, wgt - (1 << s->wp_denom[1]));
                                bw_se(w, s->wp_o[l][i][c] - 128 + ((128 * wgt) >> s->wp_denom[1])); }
                    }
                }
            }
            bw_ue(w, (uint32_t)(5 - s->max_merge));
        }
        bw_se(w, 2);                                                    /* slice_qp_delta (see write_pps) */
        if (p->deblock == 2) {
            bw_put(w, 1, 1);                                            /* deblocking_filter_override_flag */
            bw_put(w, 1, (uint32_t)s->deblock_off);
            if (!s->deblock_off) { bw_se(w, s->beta); bw_se(w, s->tc); }
        }
        if (p->sao ? (s->sao_l || s->sao_c || !s->deblock_off) : !s->deblock_off) bw_put(w, 1, (uint32_t)s->lf_across);
    }
    if (p->wpp || p->tile_cols * p->tile_rows > 1) {
        bw_ue(w, (uint32_t)n_entry);
        if (n_entry > 0) { size_t mx = 1; for (int i = 0; i < n_entry; i++) if (sizes[i] > mx) mx = sizes[i]; int len = 1;
            while (((size_t)1 << len) < mx) len++; bw_ue(w, (uint32_t)(len - 1)); for (int i = 0; i < n_entry; i++) bw_put(w, len, (uint32_t)(sizes[i] - 1)); }
    }
    bw_put(w, 1, 1); while (w->nbits) bw_put(w, 1, 0);                /* byte_alignment() */
}

/* slice-level parameters and reference picture lists (8.3.4) of a new independent slice */
static void begin_slice(Enc *e, const PicPlan *pp, const Sched *sc, int addr) {
    const HevcGenParams *p = &e->p; Rng *r = &e->rng;
    Slc *s = &e->slices[e->n_slices++]; memset(s, 0, sizeof *s); e->sl = s;
    s->addr = addr; s->type = pp->type; s->qp = p->qp;
    s->deblock_off = p->deblock == 0 || (p->deblock == 2 && rnd_n(r, 6) == 0);
    s->beta = p->deblock == 2 ? rnd_n(r, 13) - 6 : 0; s->tc = p->deblock == 2 ? rnd_n(r, 13) - 6 : 0;
    s->sao_l = p->sao && rnd_n(r, 8) != 0; s->sao_c = p->sao && rnd_n(r, 8) != 0;
    s->lf_across = rnd_n(r, 4) != 0;
    if (!(s->sao_l || s->sao_c || !s->deblock_off)) s->lf_across = 1;      /* the flag is then not in the header: inferred = the PPS flag (7.4.7.1) */
    s->max_merge = p->merge_cand;
    if (s->type == 2) return;
    s->n_ref[0] = sc->n[0]; s->n_ref[1] = s->type == 0 ? sc->n[1] : 0;
    if (p->mode == 1 && rnd_n(r, 3) == 0) { s->n_ref[0] = 1 + rnd_n(r, MIN(4, 2 * pp->n_total));
        if (s->type == 0) s->n_ref[1] = 1 + rnd_n(r, MIN(4, 2 * pp->n_total)); }
    for (int l = 0; l < (s->type == 0 ? 2 : 1); l++) {
        Pic *tmp[32]; int n = 0, want = MAX(s->n_ref[l], pp->n_total);
        while (n < want) {
            for (int i = 0; i < (l ? pp->na : pp->nb) && n < want; i++) tmp[n++] = l ? pp->after[i] : pp->before[i];
            for (int i = 0; i < (l ? pp->nb : pp->na) && n < want; i++) tmp[n++] = l ? pp->before[i] : pp->after[i];
            for (int i = 0; i < pp->nl && n < want; i++) tmp[n++] = pp->ltc[i];
        }
        s->rplm_flag[l] = p->rplm && pp->n_total > 1 && rnd_n(r, 2);
        for (int i = 0; i < s->n_ref[l]; i++) {
            int k = i;
            if (s->rplm_flag[l]) { k = rnd_n(r, pp->n_total); s->list_entry[l][i] = k; }
            s->ref[l][i] = tmp[k]; s->ref_poc[l][i] = tmp[k]->poc; s->ref_lt[l][i] = tmp[k]->is_ref == 2;
        }
    }
    s->mvd_l1_zero = s->type == 0 && rnd_n(r, 4) == 0;
    s->cabac_init = p->cabac_init == 2 ? rnd_n(r, 2) : p->cabac_init;
    s->tmvp = p->tmvp && rnd_n(r, 8) != 0;
    s->col_l0 = s->type == 0 ? rnd_n(r, 2) : 1;
    s->col_idx = rnd_n(r, s->col_l0 ? s->n_ref[0] : s->n_ref[1]);
    if (p->wp) {
        s->wp_on = 1; s->wp_denom[0] = rnd_n(r, 8); { int t = s->wp_denom[0] + rnd_n(r, 5) - 2; s->wp_denom[1] = CLIP3(0, 7, t); }
        for (int l = 0; l < 2; l++) for (int i = 0; i < s->n_ref[l]; i++) for (int c = 0; c < 3; c++) {
            int dn = s->wp_denom[c ? 1 : 0], plain = rnd_n(r, 3) == 0;
            if (c == 2 && (s->wp_w[l][i][1] == (1 << s->wp_denom[1]) && s->wp_o[l][i][1] == 0)) plain = rnd_n(r, 2);
            s->wp_w[l][i][c] = (1 << dn) + (plain ? 0 : rnd_n(r, 2 * MAX(1, (1 << dn) / 4) + 1) - MAX(1, (1 << dn) / 4));
            s->wp_o[l][i][c] = plain ? 0 : rnd_n(r, 21) - 10;
        }
    }
}

static void store_col_motion(Enc *e) {
    Pic *p = e->cur; int cw = (e->W + 15) >> 4, ch = (e->H + 15) >> 4;
    for (int y = 0; y < ch; y++) for (int x = 0; x < cw; x++) {
        int i = I4(e, x * 16, y * 16), k = y * cw + x; const Slc *s = &e->slices[e->slice_of[i]];
        p->col_intra[k] = e->pm[i] != 1; p->col[k] = e->mot[i]; p->col_lt[k] = 0;
        for (int l = 0; l < 2; l++) if ((e->mot[i].pf >> l) & 1) { p->col_poc[2 * k + l] = s->ref_poc[l][e->mot[i].ref[l]];
            p->col_lt[k] |= (uint8_t)(s->ref_lt[l][e->mot[i].ref[l]] << l); }
    }
}

static void tables_init(Enc *e) {                                       /* 6.5.1 */
    const HevcGenParams *p = &e->p; int nc = p->tile_cols, nr = p->tile_rows, *cb = e->tile_cb, *rb = e->tile_rb;
    for (int i = 0; i <= nc; i++) cb[i] = (i * e->ctb_w) / nc;
    for (int i = 0; i <= nr; i++) rb[i] = (i * e->ctb_h) / nr;
    e->tile_explicit = nc * nr > 1 && (p->seed & 2);
    if (e->tile_explicit) {                                             /* move the inner boundaries around (every tile keeps at least one CTB) */
        Rng r = { (uint64_t)p->seed * 31 + 7 };
        for (int i = 1; i < nc; i++) { int lo = cb[i - 1] + 1, hi = e->ctb_w - (nc - i); cb[i] = lo + rnd_n(&r, hi - lo + 1); }
        for (int i = 1; i < nr; i++) { int lo = rb[i - 1] + 1, hi = e->ctb_h - (nr - i); rb[i] = lo + rnd_n(&r, hi - lo + 1); }
    }
    int ts = 0;
    for (int tr = 0; tr < nr; tr++) for (int tc = 0; tc < nc; tc++)
        for (int y = rb[tr]; y < rb[tr + 1]; y++) for (int x = cb[tc]; x < cb[tc + 1]; x++) { int rs = y * e->ctb_w + x; e->rs2ts[rs] = ts; e->ts2rs[ts] = rs;
            e->tile_of[ts] = tr * nc + tc; ts++; }
}

static void encode_picture(Enc *e, BitW *out, const Sched *sched, int n_sched, int idx) {
    const HevcGenParams *p = &e->p; const Sched *sc = &sched[idx]; Rng *r = &e->rng;
    PicPlan pp; memset(&pp, 0, sizeof pp);
    pp.type = sc->type; pp.idr = sc->idr; pp.poc = sc->poc; pp.nal = sc->idr ? 19 : (sc->cra ? 21 : (sc->rasl ? 8 : (sc->is_ref ? 1 : 0)));
    if (sc->idr) for (int i = 0; i < 10; i++) e->dpb[i].used = 0;
    /* reference picture set: every earlier reference picture of this period that this or a later picture still needs */
    int period0 = idx; while (!sched[period0].idr) period0--;
    if (!sc->idr) {
        RpsSet rs; plan_rps(sched, n_sched, idx, &rs);
        pp.n_neg = rs.n_neg; pp.n_pos = rs.n_pos;
        for (int i = 0; i < rs.n_neg; i++) { pp.dneg[i] = rs.dneg[i]; pp.uneg[i] = rs.uneg[i];
            if (rs.uneg[i]) pp.before[pp.nb++] = find_poc(e, sc->poc + rs.dneg[i]); }
        for (int i = 0; i < rs.n_pos; i++) { pp.dpos[i] = rs.dpos[i]; pp.upos[i] = rs.upos[i];
            if (rs.upos[i]) pp.after[pp.na++] = find_poc(e, sc->poc + rs.dpos[i]); }
        if (sc->lt) {                                                   /* the IDR picture of the period serves as a long-term reference */
            Pic *pic = find_poc(e, sched[period0].poc);
            int max = 1 << e->poc_bits;
            pic->is_ref = 2;
            pp.ltc[pp.nl] = pic; pp.lt_poc_lsb[pp.nl] = pic->poc & (max - 1);
            pp.lt_msb[pp.nl] = (sc->poc - pic->poc) >= max / 2 || rnd_n(r, 2);
            pp.lt_cycle[pp.nl] = ((sc->poc & ~(max - 1)) - (pic->poc & ~(max - 1))) >> e->poc_bits; pp.nl++;
        }
        /* pictures that fell out of the set are no longer references */
        for (int i = 0; i < 10; i++) if (e->dpb[i].used && e->dpb[i].is_ref == 1) { int keep = 0;
            for (int k = 0; k < pp.n_neg; k++) keep |= e->dpb[i].poc == sc->poc + pp.dneg[k];
            for (int k = 0; k < pp.n_pos; k++) keep |= e->dpb[i].poc == sc->poc + pp.dpos[k]; if (!keep) e->dpb[i].is_ref = 0; }
        for (int i = 0; i < 10; i++) if (e->dpb[i].used && e->dpb[i].is_ref == 2) { int keep = 0;
            for (int k = 0; k < pp.nl; k++) keep |= pp.ltc[k] == &e->dpb[i]; if (!keep) e->dpb[i].is_ref = 0; }
        pp.n_total = pp.nb + pp.na + pp.nl;
    }
    Pic *cur = NULL;
    for (int i = 0; i < 10 && !cur; i++) if (!e->dpb[i].used || !e->dpb[i].is_ref) cur = &e->dpb[i];
    cur->used = 1; cur->is_ref = 0; cur->poc = sc->poc; cur->type = sc->type; e->cur = cur;
    make_source(e, sc->t);
    size_t n4 = (size_t)e->w4 * e->h4;
    memset(e->pm, 0, n4); memset(e->skip, 0, n4); memset(e->depth, 0, n4); memset(e->nofilt, 0, n4); memset(e->edges, 0, n4); memset(e->cbf, 0, n4);
    for (int i = 0; i < e->ctb_w * e->ctb_h; i++) e->ctb_slice[i] = -1;
    e->n_slices = 0;
    const int n_ctb = e->ctb_w * e->ctb_h, tiles = p->tile_cols * p->tile_rows > 1;
    int slice_ctus = p->slice_ctus;
    if (p->wpp && slice_ctus > 0) slice_ctus = MAX(1, (slice_ctus + e->ctb_w - 1) / e->ctb_w) * e->ctb_w;
    uint8_t wpp_st[HG_N_CTX], wpp_mps[HG_N_CTX]; int wpp_valid = 0;
    int seg_index = 0, ts = 0;
    while (ts < n_ctb) {
        /* ---- one slice segment ---- */
        int first_ts = ts, dependent = p->dep_slices && seg_index > 0 && (seg_index & 1);
        int seg_addr = e->ts2rs[ts];
        if (!dependent) { begin_slice(e, &pp, sc, seg_addr); e->seg_first_ctb = seg_addr; }
        Slc *s = e->sl;
        BitW data; memset(&data, 0, sizeof data);
        size_t marks[600]; int n_marks = 0;
        Cab *cb = &e->cab;
        int init_type = s->type == 2 ? 0 : (s->type == 1 ? (s->cabac_init ? 2 : 1) : (s->cabac_init ? 1 : 2));
        if (!dependent) { cab_init_ctx(cb, init_type, s->qp); e->first_qg = 1; e->qg_open = 0; e->qp_prev = s->qp; }
        else { e->qp_prev = e->last_cu_qp; e->first_qg = 0; e->qg_open = 0; }
        cab_start(cb, &data);
        for (;;) {
            int rs = e->ts2rs[ts], rx = rs % e->ctb_w, ry = rs / e->ctb_w, tile = e->tile_of[ts];
            int first_in_tile = ts == 0 || e->tile_of[ts - 1] != tile;
            int row_start = p->wpp && (rx == 0 || e->tile_of[e->rs2ts[rs - 1]] != tile);
            e->ctb_slice[rs] = e->seg_first_ctb; e->cur_ts = ts;
            /* the first CTB of a tile starts from initialised contexts, also at the head of a dependent slice segment (9.3.1) */
            if (first_in_tile) { if (ts != first_ts || dependent) cab_init_ctx(cb, init_type, s->qp); e->first_qg = 1; e->qg_open = 0; e->qp_prev = s->qp; }
            else if (row_start) {
                int x0 = rx << p->ctb_log2, y0 = ry << p->ctb_log2;
                if (avail(e, x0, y0, x0 + e->ctb, y0 - e->ctb) && wpp_valid) { memcpy(cb->st, wpp_st, sizeof wpp_st); memcpy(cb->mps, wpp_mps, sizeof wpp_mps);
                    }
                else if (ts != first_ts) cab_init_ctx(cb, init_type, s->qp);
                e->first_qg = 1; e->qg_open = 0; e->qp_prev = s->qp;
            }
            encode_sao(e, rs);
            encode_cqt(e, rx << p->ctb_log2, ry << p->ctb_log2, p->ctb_log2, 0);
            if (p->wpp && (rx == 1 || (rs > 1 && rx > 1 && e->tile_of[e->rs2ts[rs - 2]] != tile))) { memcpy(wpp_st, cb->st, sizeof wpp_st);
                memcpy(wpp_mps, cb->mps, sizeof wpp_mps); wpp_valid = 1; }
            ts++;
            int end = ts >= n_ctb;
            if (!end && slice_ctus > 0 && !tiles && ts - first_ts >= slice_ctus) end = 1;
            if (!end && slice_ctus > 0 && tiles && e->tile_of[ts] != e->tile_of[ts - 1]) end = 1;
            cab_term(cb, end);
            if (end) { while (data.nbits) bw_put(&data, 1, 0); break; }
            int nrs = e->ts2rs[ts];
            if ((tiles && e->tile_of[ts] != e->tile_of[ts - 1]) || (p->wpp && (nrs % e->ctb_w == 0 || e->tile_of[ts] != e->tile_of[e->rs2ts[nrs - 1]]))) {
                cab_term(cb, 1);                                        /* end_of_subset_one_bit + byte_alignment() */
                while (data.nbits) bw_put(&data, 1, 0);
                if (n_marks < 600) marks[n_marks++] = data.len;
                cab_start(cb, &data);
            }
        }
        /* entry point sizes in escaped bytes */
        size_t sizes[600]; int n_entry = n_marks;
        { BitW tmp; memset(&tmp, 0, sizeof tmp); size_t mk[601]; for (int i = 0; i < n_marks; i++) mk[i] = marks[i];
          int zeros = 0, mi = 0; size_t pos = 0, prev = 0;
          for (size_t i = 0; i < data.len; i++) { while (mi < n_marks && mk[mi] == i) { sizes[mi] = pos - prev; prev = pos; mi++; }
              if (zeros >= 2 && data.buf[i] <= 3) { pos++; zeros = 0; } pos++; zeros = data.buf[i] == 0 ? zeros + 1 : 0; }
          (void)tmp; }
        BitW nal; memset(&nal, 0, sizeof nal);
        write_slice_header(e, &nal, &pp, s, first_ts == 0, dependent, seg_addr, sizes, n_entry);
        bw_bytes(&nal, data.buf, data.len);
        write_nal(out, pp.nal, 0, nal.buf, nal.len, NULL, 0);
        free(nal.buf); free(data.buf);
        seg_index++;
    }
    (void)r;
    deblock_picture(e);
    sao_picture(e);
    store_col_motion(e);
    cur->is_ref = sc->is_ref ? 1 : 0;
    if (e->recon_buf && sc->t < e->recon_frames) {
        uint8_t *o = e->recon_buf + (size_t)sc->t * ((size_t)p->width * p->height * 3 / 2);
        for (int y = 0; y < p->height; y++) memcpy(o + (size_t)y * p->width, cur->pl[0] + (size_t)y * cur->stride[0], (size_t)p->width);
        o += (size_t)p->width * p->height;
        for (int c = 1; c < 3; c++) { for (int y = 0; y < p->height / 2; y++) memcpy(o + (size_t)y * (p->width / 2), cur->pl[c] + (size_t)y * cur->stride[c],
            (size_t)(p->width / 2)); o += (size_t)(p->width / 2) * (p->height / 2); }
    }
}

/* library entry: returns a malloc'ed Annex-B stream; the reconstruction (display order, cropped, I420) goes to recon_path */
int hevcgen_generate(const HevcGenParams *gp, uint8_t **out, size_t *out_len, const char *recon_path) {
    Enc *e = (Enc *)calloc(1, sizeof(Enc));
    e->p = *gp; HevcGenParams *p = &e->p;
    if (p->width < 16 || p->height < 16 || (p->width & 1) || (p->height & 1) || p->frames < 1) { free(e); return -1; }
    p->ctb_log2 = CLIP3(4, 6, p->ctb_log2 ? p->ctb_log2 : 6); p->min_cb_log2 = CLIP3(3, p->ctb_log2, p->min_cb_log2 ? p->min_cb_log2 : 3);
    p->min_tb_log2 = CLIP3(2, p->min_cb_log2 - 1, p->min_tb_log2 ? p->min_tb_log2 : 2);
    p->max_tb_log2 = CLIP3(p->min_tb_log2, MIN(5, p->ctb_log2), p->max_tb_log2 ? p->max_tb_log2 : 5);
    p->depth_inter = CLIP3(0, p->ctb_log2 - p->min_tb_log2, p->depth_inter); p->depth_intra = CLIP3(0, p->ctb_log2 - p->min_tb_log2, p->depth_intra);
    p->qp = CLIP3(4, 48, p->qp ? p->qp : 32); if (p->intra_period < 1) p->intra_period = 32;
    if (!(p->gop == 0 || p->gop == 8 || (p->gop >= 1 && p->gop <= 3))) p->gop = 0;
    p->num_ref = CLIP3(1, 4, p->num_ref ? p->num_ref : 1); if (p->gop == 8 && p->num_ref > 2) p->num_ref = 2;
    if (p->gop >= 1 && p->gop <= 3 && p->num_ref > 2) p->num_ref = 2;
    p->merge_cand = CLIP3(1, 5, p->merge_cand ? p->merge_cand : 5); p->par_mrg = CLIP3(2, p->ctb_log2, p->par_mrg ? p->par_mrg : 2);
    p->tile_cols = MAX(1, p->tile_cols); p->tile_rows = MAX(1, p->tile_rows); if (p->search < 1) p->search = 4;
    /* the slice table holds 512 entries (dependent slice segments share their parent's): keep independent slices per picture below that */
    if (p->slice_ctus > 0) {
        const int cs = 1 << p->ctb_log2, n_ctbs = ((p->width + cs - 1) / cs) * ((p->height + cs - 1) / cs);
        if ((n_ctbs + p->slice_ctus - 1) / p->slice_ctus > 400) p->slice_ctus = (n_ctbs + 399) / 400;
    }
    p->dqp = CLIP3(0, 1 + MIN(3, p->ctb_log2 - p->min_cb_log2), p->dqp); p->scaling = CLIP3(0, 3, p->scaling); p->pcm = CLIP3(0, 3, p->pcm);
    p->deblock = CLIP3(0, 2, p->deblock);
    p->cb_qp_off = CLIP3(-12, 12, p->cb_qp_off); p->cr_qp_off = CLIP3(-12, 12, p->cr_qp_off);
    if (p->gop) p->lt_ref = 0;
    int mcb = 1 << p->min_cb_log2;
    e->W = (p->width + mcb - 1) & ~(mcb - 1); e->H = (p->height + mcb - 1) & ~(mcb - 1);
    e->ctb = 1 << p->ctb_log2; e->ctb_w = (e->W + e->ctb - 1) >> p->ctb_log2; e->ctb_h = (e->H + e->ctb - 1) >> p->ctb_log2; e->w4 = e->W / 4; e->h4 = e->H / 4;
    p->tile_cols = MIN(p->tile_cols, MIN(e->ctb_w, 20)); p->tile_rows = MIN(p->tile_rows, MIN(e->ctb_h, 22));
    if (p->tile_cols * p->tile_rows > 1) p->wpp = 0;
    e->lf_across_tiles = !(p->seed & 1);
    e->poc_bits = 5 + (p->seed & 3); if (p->lt_ref) e->poc_bits = 4 + (p->seed & 1) * 4;
    if (p->gop == 8 && e->poc_bits < 5) e->poc_bits = 5;
    e->rng.s = (uint64_t)p->seed * 0x9E3779B97F4A7C15ull + 777;
    basis_init();
    pic_alloc(e, &e->src); for (int i = 0; i < 10; i++) pic_alloc(e, &e->dpb[i]);
    size_t n4 = (size_t)e->w4 * e->h4, nc = (size_t)e->ctb_w * e->ctb_h;
    e->pm = calloc(n4, 1); e->skip = calloc(n4, 1); e->depth = calloc(n4, 1); e->ipm = calloc(n4, 1); e->nofilt = calloc(n4, 1); e->edges = calloc(n4, 1);
    e->cbf = calloc(n4, 1);
    e->qpmap = calloc(n4, 1); e->mot = calloc(n4, sizeof(Mot)); e->slice_of = calloc(n4, sizeof(int16_t));
    e->ctb_slice = calloc(nc, sizeof(int)); e->rs2ts = calloc(nc, sizeof(int)); e->ts2rs = calloc(nc, sizeof(int)); e->tile_of = calloc(nc, sizeof(int));
    e->sao = calloc(nc, sizeof(Sao));
    for (int c = 0; c < 3; c++) e->dbk[c] = malloc((size_t)(e->W >> (c ? 1 : 0)) * (size_t)(e->H >> (c ? 1 : 0)));
    tables_init(e);
    make_texture(e);
    if (recon_path) { e->recon = fopen(recon_path, "wb"); e->recon_frames = p->frames;
        e->recon_buf = calloc((size_t)p->frames, (size_t)p->width * p->height * 3 / 2); }
    Sched *sched = calloc((size_t)p->frames + 16, sizeof(Sched)); int n_sched = 0;
    plan_sequence(p, sched, &n_sched);
    int reorder = p->gop == 8 ? 3 : (p->gop ? 1 : 0), keep_max = p->gop == 8 ? 5 : p->num_ref + (p->gop ? 1 : 0) + (p->lt_ref ? 1 : 0);
    int max_dpb = MIN(16, keep_max + reorder + 1);
    BitW outw; memset(&outw, 0, sizeof outw);
    lists_default(e); lists_expand(e); e->sf_on = p->scaling != 0;
    if (p->rps_sps && !p->lt_ref) {
        /* the distinct reference picture sets of the plan go into the SPS (some are left out on purpose) */
        for (int i = 0; i < n_sched && e->n_sps_sets < 64; i++) {
            if (sched[i].idr) continue;
            RpsSet t; plan_rps(sched, n_sched, i, &t);
            int known = 0; for (int k = 0; k < e->n_sps_sets; k++) known |= rps_equal(&e->sps_sets[k], &t);
            if (!known && rnd_n(&e->rng, 4) != 0) e->sps_sets[e->n_sps_sets++] = t;
        }
    }
    write_vps(e, &outw, max_dpb, reorder);
    write_sps(e, &outw, max_dpb, reorder);
    write_pps(e, &outw);
    size_t ps_len = outw.len;
    if (p->scaling >= 2) lists_expand(e);
    for (int i = 0; i < n_sched; i++) {
        if (sched[i].cra) { uint8_t *copy = (uint8_t *)malloc(ps_len); memcpy(copy, outw.buf, ps_len); bw_bytes(&outw, copy, ps_len); free(copy); }
            /* parameter sets again: decoding may start here */
        encode_picture(e, &outw, sched, n_sched, i);
    }
    if (e->recon) { fwrite(e->recon_buf, 1, (size_t)p->frames * ((size_t)p->width * p->height * 3 / 2), e->recon); fclose(e->recon); }
    *out = outw.buf; *out_len = outw.len;
    free(sched);
    /* everything the encoder allocated (a sweep of 50,000 streams in one process kept 0.7 MB of it per call) */
    {
        Pic *pics[11]; pics[0] = &e->src; for (int i = 0; i < 10; i++) pics[1 + i] = &e->dpb[i];
        for (int i = 0; i < 11; i++) { for (int c = 0; c < 3; c++) free(pics[i]->pl[c]); free(pics[i]->col); free(pics[i]->col_poc); free(pics[i]->col_lt); free(pics[i]->col_intra); }
        free(e->pm); free(e->skip); free(e->depth); free(e->ipm); free(e->nofilt); free(e->edges); free(e->cbf); free(e->qpmap); free(e->mot); free(e->slice_of);
        free(e->ctb_slice); free(e->rs2ts); free(e->ts2rs); free(e->tile_of); free(e->sao);
        for (int c = 0; c < 3; c++) free(e->dbk[c]);
        free(e->tex); free(e->recon_buf); free(e);
    }
    return 0;
}

void hevcgen_free(void *p) { free(p); }

#ifndef HEVCGEN_LIB
int main(int argc, char **argv) {
    HevcGenParams p; memset(&p, 0, sizeof p);
    p.width = 176; p.height = 144; p.frames = 8; p.qp = 32; p.seed = 1; p.intra_period = 32; p.deblock = 1; p.sao = 1; p.tmvp = 1; p.amp = 1;
    p.strong_intra = 1; p.depth_inter = 2; p.depth_intra = 2;
    const char *outp = NULL, *recon = NULL;
    struct { const char *name; int *v; } opts[] = { {"--width", &p.width}, {"--height", &p.height}, {"--frames", &p.frames}, {"--qp", &p.qp}, {"--seed",
        &p.seed}, {"--intra-period", &p.intra_period},
        {"--gop", &p.gop}, {"--refs", &p.num_ref}, {"--ctb", &p.ctb_log2}, {"--min-cb", &p.min_cb_log2}, {"--max-tb", &p.max_tb_log2}, {"--min-tb",
            &p.min_tb_log2}, {"--depth-inter", &p.depth_inter},
        {"--depth-intra", &p.depth_intra}, {"--mode", &p.mode}, {"--amp", &p.amp}, {"--sao", &p.sao}, {"--deblock", &p.deblock}, {"--tskip", &p.tskip},
            {"--sdh", &p.sdh}, {"--dqp", &p.dqp}, {"--pcm", &p.pcm},
        {"--bypass", &p.bypass}, {"--cip", &p.cip}, {"--strong-intra", &p.strong_intra}, {"--tmvp", &p.tmvp}, {"--wp", &p.wp}, {"--rplm", &p.rplm},
            {"--lt-ref", &p.lt_ref}, {"--scaling", &p.scaling},
        {"--wpp", &p.wpp}, {"--tile-cols", &p.tile_cols}, {"--tile-rows", &p.tile_rows}, {"--slice-ctus", &p.slice_ctus}, {"--dep-slices", &p.dep_slices},
            {"--merge-cand", &p.merge_cand},
        {"--cabac-init", &p.cabac_init}, {"--par-mrg", &p.par_mrg}, {"--cb-qp-off", &p.cb_qp_off}, {"--cr-qp-off", &p.cr_qp_off}, {"--search", &p.search},
            {"--rps-sps", &p.rps_sps}, {"--open-gop", &p.open_gop}, {"--vui-fps", &p.vui_fps} };
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-o") && i + 1 < argc) { outp = argv[++i]; continue; }
        if (!strcmp(argv[i], "--recon") && i + 1 < argc) { recon = argv[++i]; continue; }
        int ok = 0;
        for (size_t k = 0; k < sizeof opts / sizeof opts[0]; k++) if (!strcmp(argv[i], opts[k].name) && i + 1 < argc) {
            *opts[k].v = (int)strtol(argv[++i], NULL, 0); ok = 1; break; }
        if (!ok) { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (!outp) { fprintf(stderr, "usage: hevcgen [--width W --height H --frames N --qp Q --seed S --gop 0|1|2|3|8 ...] -o out.h265 [--recon recon.yuv]\n");
        return 2; }
    uint8_t *buf; size_t len;
    if (hevcgen_generate(&p, &buf, &len, recon) < 0) { fprintf(stderr, "bad parameters\n"); return 1; }
    FILE *f = fopen(outp, "wb"); fwrite(buf, 1, len, f); fclose(f);
    fprintf(stderr, "wrote %zu bytes, %d frames (%.1f kbit/frame)\n", len, p.frames, len * 8.0 / 1000 / p.frames);
    free(buf);
    return 0;
}
#endif
