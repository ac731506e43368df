/*
 * tools/h264gen.c -- seeded synthetic H.264 Annex-B stream generator.
 *
 * The reference repository ships no bitstreams, fixtures or golden YUV
 *  and this image has no encoder, so every test and bench
 * input is produced here: a small closed-loop Constrained-Baseline encoder
 * (CAVLC, I/P, I_PCM, Intra16x16 / Intra4x4, P16x16/16x8/8x16/8x8 with
 * 8x4/4x8/4x4 sub-partitions, quarter-pel MVs that may point outside the
 * picture, multiple reference frames, multiple slices, in-loop deblocking)
 * with its OWN reconstruction loop.  The reconstruction it writes with
 * --recon must equal what any conforming decoder outputs for the stream, which
 * gives the CPU oracle and the HIP decoder a second, independently written
 * code path to agree with (tests/test_oracle_vs_generator.py).
 *
 * Not derived from the reference (which contains no codec arithmetic).
 * Content is procedural and integer-only so that streams are bit-identical on
 * every machine:  seed = 0x4A4D0000 + config_id*256 + stream_id.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PAD 48
#define CLIP3(lo, hi, v) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))
#define CLIP1(v) CLIP3(0, 255, v)
#define ABS(v) ((v) < 0 ? -(v) : (v))
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))

typedef struct {
    int width, height;          /* display size (cropped)                     */
    int frames, qp, gop, seed;
    int mode;                   /* 0 = "real" (SAD decisions), 1 = fuzz       */
    int deblock;                /* 1 = on (idc 0), 0 = off (idc 1), 2 = idc 2 */
    int num_ref;                /* max_num_ref_frames 1..4                    */
    int slices;                 /* slices per picture (split by MB rows)      */
    int pcm_only;               /* every MB I_PCM (known-answer streams)      */
    int poc_type;               /* 0, 1 or 2 (1: P-only streams; expected-delta cycle, offsets and delta_pic_order_cnt[] drawn from the seed) */
    int nonref_period;          /* >0: every n-th P picture is non-reference  */
    int alpha_off, beta_off;    /* slice_alpha_c0_offset_div2, beta_offset_div2 */
    int chroma_qp_off;
    int level_idc;
    int cip;                    /* constrained_intra_pred_flag                */
    int search;                 /* integer search range                       */
    int cabac;                  /* entropy_coding_mode_flag                   */
    int cabac_idc;              /* cabac_init_idc 0..2                        */
    int t8x8;                   /* transform_8x8_mode_flag: 8x8 transform + Intra8x8 (High profile) */
    int bframes;                /* B pictures between anchors (0..3); forces pic_order_cnt_type 0 */
    int direct_temporal;        /* 0: direct_spatial_mv_pred_flag = 1, 1: temporal direct */
    int wp;                     /* 0 none, 1 explicit weights (P and B), 2 implicit (B)          */
    int dinf8;                  /* direct_8x8_inference_flag (default 1 when 0 is passed with bframes == 0) */
    int scaling;                /* 0 flat, 1 scaling lists in the SPS, 2 in the PPS (forces High profile)             */
    int rplm;                   /* 1: random ref_pic_list_modification() in P / B slices (fuzz)                       */
    int mmco;                   /* 1: random memory_management_control_operations, long-term references (P-only streams); 2: also operation 5 now and
                                   then (all references dropped, frame_num and picture order count restart: 7.4.3, 8.2.1) */
    int nc_corner;              /* 1: Intra4x4 modes 4-6 may be chosen although p[-1,-1] is unavailable (NON-CONFORMING; probes the decoders'
                                   common convention "unavailable samples count as 128", which constrained_intra_pred exposes)   */
    int no_intra;               /* 1: fuzz mode codes no intra macroblocks in P / B pictures and no non-IDR I pictures (the random numbers are
                                   still drawn, so the rest of the stream's decisions do not shift)                              */
    int fmo0;                   /* 1: frame_mbs_only_flag = 0 without MBAFF, every picture a FRAME picture (field_pic_flag = 0): an interlace-capable
                                   stream that happens to be coded progressively.  Needs an even number of macroblock rows; Main profile at least */
    int poc_bottom;             /* 1: bottom_field_pic_order_in_frame_present_flag = 1 with random delta_pic_order_cnt_bottom / delta_pic_order_cnt[1] in -1..1
                                   (PicOrderCnt of a frame = Min(top, bottom), 8.2.1) */
    int paff;                   /* picture-adaptive frame / field coding (implies fmo0): 1 = every I / P picture is coded either as a frame or as two field
                                   pictures (first field of either parity), drawn per picture (P-only streams); 2 = every picture as two fields (B
                                   pictures allowed).  With
                                   CABAC the 8x8 transform is switched off (the contexts 436..459 of field-coded 8x8 blocks are not pinned, SPEC_AUDIT.md) */
    int gaps;                   /* 1: gaps_in_frame_num_value_allowed_flag = 1 and, now and then, one or two frame_num values are skipped before a P picture
                                   (P-only streams): the decoder must infer frames that were never sent (8.2.5.2) -- they pass through the sliding window,
                                   push older pictures out and sit in the reference lists, where nothing may predict from them */
    int redundant;              /* 1: redundant_pic_cnt_present_flag = 1; behind the slices of a picture now and then a slice of a REDUNDANT coded picture
                                   (redundant_pic_cnt 1 or 2: the header of the picture's first slice, then bits that are not slice data) -- a decoder
                                   decodes the primary picture and must leave these alone (Baseline tool) */
    int vui_fps;                /* > 0: the SPS carries VUI timing_info (num_units_in_tick 1, time_scale 2 * vui_fps, fixed_frame_rate_flag 1) and an
                                   aspect ratio -- nothing in it influences decoding; jm_intel_get_stream_info reports the frame rate from it */
} GenParams;

/* ------------------------------ RNG --------------------------------------- */
typedef struct { uint64_t s; } Rng;
static uint32_t rnd(Rng *r) { r->s = r->s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(r->s >> 33); }
static int rnd_n(Rng *r, int n) { return (int)(rnd(r) % (uint32_t)n); }

/* ------------------------------ bit writer -------------------------------- */
typedef struct { uint8_t *buf; size_t cap, len; uint32_t cur; int nbits; } BitW;
static void bw_reserve(BitW *w, size_t extra) {
    if (w->len + extra > w->cap) { w->cap = (w->len + extra) * 2 + 1024; w->buf = (uint8_t *)realloc(w->buf, w->cap); }
}
static void bw_put(BitW *w, int n, uint32_t v) {
    for (int i = n - 1; i >= 0; i--) {
        w->cur = (w->cur << 1) | ((v >> i) & 1);
        if (++w->nbits == 8) { bw_reserve(w, 1); w->buf[w->len++] = (uint8_t)w->cur; w->cur = 0; w->nbits = 0; }
    }
}
static void bw_ue(BitW *w, uint32_t v) {
    uint32_t x = v + 1; int n = 0;
    while ((x >> n) > 1) n++;
    bw_put(w, n, 0); bw_put(w, n + 1, x);
}
static void bw_se(BitW *w, int v) { bw_ue(w, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
static void bw_te(BitW *w, int range_max, int v) { if (range_max > 1) bw_ue(w, v); else bw_put(w, 1, !v); }
static void bw_trailing(BitW *w) { bw_put(w, 1, 1); while (w->nbits) bw_put(w, 1, 0); }
static int bw_bitpos(const BitW *w) { return (int)w->len * 8 + w->nbits; }
static int bw_bit_at(const BitW *w, int pos) {                      /* bit `pos` of what has been written so far */
    if (pos < (int)w->len * 8) return (w->buf[pos >> 3] >> (7 - (pos & 7))) & 1;
    return (int)(w->cur >> (w->nbits - 1 - (pos - (int)w->len * 8))) & 1;
}

/* ---------------- CABAC encoder (9.3.4) ------------------------------------- */
#include "../oracle/orc_cabac_tables.h"     /* context init (m,n), rangeTabLPS, transIdxLPS: data tables shared with the oracle */
typedef struct { uint8_t st[ORC_CABAC_N_CTX]; uint32_t low, range; int outstanding, first; BitW *w; } CabEnc;
static void cab_init_ctx(CabEnc *c, int table, int qp) {
    for (int i = 0; i < ORC_CABAC_N_CTX; i++) {
        int pre = CLIP3(1, 126, ((orc_cabac_init_mn[table][i][0] * CLIP3(0, 51, qp)) >> 4) + orc_cabac_init_mn[table][i][1]);
        c->st[i] = pre <= 63 ? (uint8_t)((63 - pre) << 1) : (uint8_t)(((pre - 64) << 1) | 1);
    }
}
static void cab_start(CabEnc *c, BitW *w) { c->w = w; c->low = 0; c->range = 510; c->outstanding = 0; c->first = 1; }
static void cab_put(CabEnc *c, int b) {
    if (c->first) c->first = 0; else bw_put(c->w, 1, (uint32_t)b);
    while (c->outstanding > 0) { bw_put(c->w, 1, (uint32_t)(1 - b)); c->outstanding--; }
}
static void cab_renorm(CabEnc *c) {
    while (c->range < 256) {
        if (c->low < 256) cab_put(c, 0);
        else if (c->low >= 512) { c->low -= 512; cab_put(c, 1); }
        else { c->low -= 256; c->outstanding++; }
        c->range <<= 1; c->low <<= 1;
    }
}
static void cab_enc(CabEnc *c, int ctx, int bin) {
    uint32_t st = c->st[ctx] >> 1, mps = c->st[ctx] & 1, lps = orc_cabac_range_lps[st][(c->range >> 6) & 3];
    c->range -= lps;
    if ((uint32_t)bin != mps) { c->low += c->range; c->range = lps; if (st == 0) mps ^= 1; st = orc_cabac_trans_lps[st]; }
    else if (st < 62) st++;
    c->st[ctx] = (uint8_t)((st << 1) | mps);
    cab_renorm(c);
}
static void cab_byp(CabEnc *c, int bin) {
    c->low <<= 1;
    if (bin) c->low += c->range;
    if (c->low >= 1024) { cab_put(c, 1); c->low -= 1024; }
    else if (c->low < 512) cab_put(c, 0);
    else { c->low -= 512; c->outstanding++; }
}
static void cab_term(CabEnc *c, int bin) {
    c->range -= 2;
    if (bin) {
        c->low += c->range;
        c->range = 2; cab_renorm(c);                          /* EncodeFlush */
        cab_put(c, (int)((c->low >> 9) & 1));
        bw_put(c->w, 2, ((c->low >> 7) & 3) | 1);
    } else cab_renorm(c);
}
static void cab_ueg(CabEnc *c, int v, int k) {               /* Exp-Golomb order k suffix, bypass coded */
    while (v >= (1 << k)) { cab_byp(c, 1); v -= 1 << k; k++; }
    cab_byp(c, 0);
    while (k--) cab_byp(c, (v >> k) & 1);
}

typedef struct { uint8_t *buf; size_t cap, len; } Out;
static void out_nal(Out *o, int ref_idc, int type, const BitW *w, int long_sc) {
    size_t need = w->len * 3 / 2 + 16;
    if (o->len + need > o->cap) { o->cap = (o->len + need) * 2; o->buf = (uint8_t *)realloc(o->buf, o->cap); }
    uint8_t *p = o->buf + o->len;
    if (long_sc) *p++ = 0;
    *p++ = 0; *p++ = 0; *p++ = 1;
    *p++ = (uint8_t)((ref_idc << 5) | type);
    int zeros = 0;
    for (size_t i = 0; i < w->len; i++) {
        uint8_t b = w->buf[i];
        if (zeros >= 2 && b <= 3) { *p++ = 3; zeros = 0; }
        *p++ = b;
        zeros = b == 0 ? zeros + 1 : 0;
    }
    o->len = (size_t)(p - o->buf);
}

/* ------------------------------ tables (own copy) -------------------------- */
static const uint8_t ct_len[4][68] = {
{ 1,0,0,0, 6,2,0,0, 8,6,3,0, 9,8,7,5, 10,9,8,6, 11,10,9,7, 13,11,10,8, 13,13,11,9, 13,13,13,10,
 14,14,13,11, 14,14,14,13, 15,15,14,14, 15,15,15,14, 16,15,15,15, 16,16,16,15, 16,16,16,16, 16,16,16,16 },
{ 2,0,0,0, 6,2,0,0, 6,5,3,0, 7,6,6,4, 8,6,6,4, 8,7,7,5, 9,8,8,6, 11,9,9,6, 11,11,11,7,
 12,11,11,9, 12,12,12,11, 12,12,12,11, 13,13,13,12, 13,13,13,13, 13,14,13,13, 14,14,14,13, 14,14,14,14 },
{ 4,0,0,0, 6,4,0,0, 6,5,4,0, 6,5,5,4, 7,5,5,4, 7,5,5,4, 7,6,6,4, 7,6,6,4, 8,7,7,5,
 8,8,7,6, 9,8,8,7, 9,9,8,8, 9,9,9,8, 10,9,9,9, 10,10,10,10, 10,10,10,10, 10,10,10,10 },
{ 6,0,0,0, 6,6,0,0, 6,6,6,0, 6,6,6,6, 6,6,6,6, 6,6,6,6, 6,6,6,6, 6,6,6,6, 6,6,6,6,
 6,6,6,6, 6,6,6,6, 6,6,6,6, 6,6,6,6, 6,6,6,6, 6,6,6,6, 6,6,6,6, 6,6,6,6 } };
static const uint8_t ct_bits[4][68] = {
{ 1,0,0,0, 5,1,0,0, 7,4,1,0, 7,6,5,3, 7,6,5,3, 7,6,5,4, 15,6,5,4, 11,14,5,4, 8,10,13,4,
 15,14,9,4, 11,10,13,12, 15,14,9,12, 11,10,13,8, 15,1,9,12, 11,14,13,8, 7,10,9,12, 4,6,5,8 },
{ 3,0,0,0, 11,2,0,0, 7,7,3,0, 7,10,9,5, 7,6,5,4, 4,6,5,6, 7,6,5,8, 15,6,5,4, 11,14,13,4,
 15,10,9,4, 11,14,13,12, 8,10,9,8, 15,14,13,12, 11,10,9,12, 7,11,6,8, 9,8,10,1, 7,6,5,4 },
{ 15,0,0,0, 15,14,0,0, 11,15,13,0, 8,12,14,12, 15,10,11,11, 11,8,9,10, 9,14,13,9, 8,10,9,8, 15,14,13,13,
 11,14,10,12, 15,10,13,12, 11,14,9,12, 8,10,13,8, 13,7,9,12, 9,12,11,10, 5,8,7,6, 1,4,3,2 },
{ 3,0,0,0, 0,1,0,0, 4,5,6,0, 8,9,10,11, 12,13,14,15, 16,17,18,19, 20,21,22,23, 24,25,26,27, 28,29,30,31,
 32,33,34,35, 36,37,38,39, 40,41,42,43, 44,45,46,47, 48,49,50,51, 52,53,54,55, 56,57,58,59, 60,61,62,63 } };
static const uint8_t cdc_len[20] = { 2,0,0,0, 6,1,0,0, 6,6,3,0, 6,7,7,6, 6,8,8,7 };
static const uint8_t cdc_bits[20] = { 1,0,0,0, 7,1,0,0, 4,6,1,0, 3,3,2,5, 2,3,2,0 };
static const uint8_t tz_len[15][16] = {
 {1,3,3,4,4,5,5,6,6,7,7,8,8,9,9,9},{3,3,3,3,3,4,4,4,4,5,5,6,6,6,6,0},{4,3,3,3,4,4,3,3,4,5,5,6,5,6,0,0},
 {5,3,4,4,3,3,3,4,3,4,5,5,5,0,0,0},{4,4,4,3,3,3,3,3,4,5,4,5,0,0,0,0},{6,5,3,3,3,3,3,3,4,3,6,0,0,0,0,0},
 {6,5,3,3,3,2,3,4,3,6,0,0,0,0,0,0},{6,4,5,3,2,2,3,3,6,0,0,0,0,0,0,0},{6,6,4,2,2,3,2,5,0,0,0,0,0,0,0,0},
 {5,5,3,2,2,2,4,0,0,0,0,0,0,0,0,0},{4,4,3,3,1,3,0,0,0,0,0,0,0,0,0,0},{4,4,2,1,3,0,0,0,0,0,0,0,0,0,0,0},
 {3,3,1,2,0,0,0,0,0,0,0,0,0,0,0,0},{2,2,1,0,0,0,0,0,0,0,0,0,0,0,0,0},{1,1,0,0,0,0,0,0,0,0,0,0,0,0,0,0} };
static const uint8_t tz_bits[15][16] = {
 {1,3,2,3,2,3,2,3,2,3,2,3,2,3,2,1},{7,6,5,4,3,5,4,3,2,3,2,3,2,1,0,0},{5,7,6,5,4,3,4,3,2,3,2,1,1,0,0,0},
 {3,7,5,4,6,5,4,3,3,2,2,1,0,0,0,0},{5,4,3,7,6,5,4,3,2,1,1,0,0,0,0,0},{1,1,7,6,5,4,3,2,1,1,0,0,0,0,0,0},
 {1,1,5,4,3,3,2,1,1,0,0,0,0,0,0,0},{1,1,1,3,3,2,2,1,0,0,0,0,0,0,0,0},{1,0,1,3,2,1,1,1,0,0,0,0,0,0,0,0},
 {1,0,1,3,2,1,1,0,0,0,0,0,0,0,0,0},{0,1,1,2,1,3,0,0,0,0,0,0,0,0,0,0},{0,1,1,1,1,0,0,0,0,0,0,0,0,0,0,0},
 {0,1,1,1,0,0,0,0,0,0,0,0,0,0,0,0},{0,1,1,0,0,0,0,0,0,0,0,0,0,0,0,0},{0,1,0,0,0,0,0,0,0,0,0,0,0,0,0,0} };
static const uint8_t ctz_len[3][4] = { {1,2,3,3},{1,2,2,0},{1,1,0,0} };
static const uint8_t ctz_bits[3][4] = { {1,1,1,0},{1,1,0,0},{1,0,0,0} };
static const uint8_t rb_len[7][15] = {
 {1,1},{1,2,2},{2,2,2,2},{2,2,2,3,3},{2,2,3,3,3,3},{2,3,3,3,3,3,3},{3,3,3,3,3,3,3,4,5,6,7,8,9,10,11} };
static const uint8_t rb_bits[7][15] = {
 {1,0},{1,1,0},{3,2,1,0},{3,2,1,1,0},{3,2,3,2,1,0},{3,0,1,3,2,5,4},{7,6,5,4,3,2,1,1,1,1,1,1,1,1,1} };
static const uint8_t cbp_intra_tab[48] = {
    47,31,15,0,23,27,29,30,7,11,13,14,39,43,45,46,16,3,5,10,12,19,21,26,28,35,37,42,44,1,2,4,8,17,18,20,24,6,9,22,25,32,33,34,36,40,38,41 };
static const uint8_t cbp_inter_tab[48] = {
    0,16,1,2,4,8,32,3,5,10,12,15,47,7,11,13,14,6,9,31,35,37,42,44,33,34,36,40,39,43,45,46,17,18,20,24,19,21,26,28,23,27,29,30,22,25,38,41 };
static const uint8_t zz4[16] = {0,1,4,8,5,2,3,6,9,12,13,10,7,11,14,15};
/* field scan (Table 8-2, raster index = 4 * y + x): down the columns first */
static const uint8_t fs4[16] = {0,4,1,8,12,5,9,13,2,6,10,14,3,7,11,15};
static const uint8_t qpc_tab[22] = {29,30,31,32,32,33,34,34,35,35,36,36,37,37,37,38,38,38,39,39,39,39};
static const int norm4[6][3] = { {10,16,13},{11,18,14},{13,20,16},{14,23,18},{16,25,20},{18,29,23} };
static const int quant_mf[6][3] = { {13107,5243,8066},{11916,4660,7490},{10082,4194,6554},{9362,3647,5825},{8192,3355,5243},{7282,2893,4559} };
static const uint8_t alpha_tab[52] = {
    0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,4,4,5,6,7,8,9,10,12,13,15,17,20,22,25,28,32,36,40,45,50,56,63,71,80,90,101,113,127,144,162,182,203,226,255,255 };
static const uint8_t beta_tab[52] = {
    0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,2,2,3,3,3,3,4,4,4,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13,14,14,15,15,16,16,17,17,18,18 };
static const uint8_t tc0_tab[52][3] = {
 {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},
 {0,0,1},{0,0,1},{0,0,1},{0,0,1},{0,1,1},{0,1,1},{1,1,1},{1,1,1},{1,1,1},{1,1,1},{1,1,2},{1,1,2},{1,1,2},{1,1,2},{1,2,3},{1,2,3},
 {2,2,3},{2,2,4},{2,3,4},{2,3,4},{3,3,5},{3,4,6},{3,4,6},{4,5,7},{4,5,8},{4,6,9},{5,7,10},{6,8,11},{6,8,13},{7,10,14},{8,11,16},
 {9,12,18},{10,13,20},{11,15,23},{13,17,25} };

/* ------------------------------ frames ------------------------------------- */
typedef struct FrameS {
    uint8_t *y, *u, *v;            /* origin pointers inside padded buffers      */
    uint8_t *by, *bu, *bv;
    int sy, sc;                    /* strides                                    */
    uint8_t *hb, *hh, *hj;         /* luma half-pel planes (origin pointers)     */
    uint8_t *bhb, *bhh, *bhj;
    int frame_num, poc, id;
    int is_long, lt_idx;           /* long-term reference with LongTermFrameIdx lt_idx */
    void *mf;                      /* MbE[] motion field of the picture (colocated data for direct prediction) */
    /* PAFF streams: every frame store also keeps its two fields as pictures of half the height (own padding, own half-sample planes): field
       pictures are coded exactly like frames of that size, against reference FIELDS; frame pictures against the woven frames */
    struct FrameS *fld;            /* [2]: top, bottom */
    uint8_t fmark[2];              /* marking of each field of the store: 0 not a reference, 1 short-term, 2 long-term */
    int fpoc[2];                   /* TopFieldOrderCnt, BottomFieldOrderCnt */
    int parity;                    /* of a field picture: 0 top, 1 bottom */
    int nonexist;                  /* a frame the decoder has to infer from a gap in frame_num (8.2.5.2): a short-term reference with no samples */
    int coded_fields;              /* the store was coded as two field pictures (its motion is kept per field, fld[].mf), not as a frame picture (mf) */
    void *store_mf; int store_fields, store_id;   /* of a field: the frame picture's motion, coding and id of its store (refreshed when lists are built) */
} Frame;

typedef struct {
    int8_t ref[4]; int16_t mv[16][2]; int refid[4];
    uint8_t tc[24]; uint8_t intra, pcm, i16, qp, qpc[2]; uint8_t i4[16];
    int slice; uint8_t skip;
    int8_t dis_db, a_off, b_off;
    int8_t ref1[4]; int16_t mv1[16][2]; int refid1[4]; uint8_t mvd1[16][2]; uint8_t direct8, bdirect16;   /* list 1 / direct prediction (B) */
    uint8_t cbp, cmode, t8; uint32_t cbf; uint8_t mvd[16][2]; uint16_t nzmask;   /* CABAC context state / 8x8 transform / deblock non-zero map */
} MbE;

typedef struct {
    GenParams p;
    int W, H, mbw, mbh;            /* coded size                                 */
    Frame src, cur, refs[5]; int nrefs;
    MbE *mbs;
    Rng rng;
    BitW bw; Out out;
    int frame_num, idr_id, log2_max_fn, poc_lsb_bits;
    int slice_id, slice_type, qp_run;
    Frame *list0[16]; int nlist0;
    Frame *list1[16]; int nlist1; int cur_poc;
    int pending_long_idx;                 /* PAFF: LongTermFrameIdx the first field of the current frame took by operation 6 (-1: none): the second follows */
    Frame fsrc; int cur_store_id;         /* PAFF: the source field being coded; id of the frame store the first field of the current frame opened */
    int field, second;                    /* the picture being coded: 0 frame, 1 top field, 2 bottom field; it is the second field of its frame */
    const uint8_t *scan4, *scan8;         /* zig-zag or field scan (8.5.6 / 8.5.7) */
    int after_op5;                        /* the previous picture carried operation 5 */
    uint8_t hdr_copy[256]; int hdr_bits, hdr_rpc;     /* redundant pictures: header bits of the current picture's first slice, position of redundant_pic_cnt */
    int usable[16], n_usable;             /* the entries of list 0 that exist (gaps in frame_num leave entries nothing may predict from) */
    int *pocs; /* PicOrderCnt of every picture by display index as the ENCODER means it (after operation 5: 0); see h264gen_last_pocs */
    int poc_base;                         /* display index at which the picture order count restarted (IDR picture, or a picture with operation 5) */
    int cur_top, delta_bottom, delta0;    /* TopFieldOrderCnt of the current picture; delta_pic_order_cnt_bottom / [1]; delta_pic_order_cnt[0] (type 1) */
    int t1_cycle, t1_ref[3], t1_nonref, t1_t2b;   /* pic_order_cnt_type 1: cycle of expected deltas, offset_for_non_ref_pic, offset_for_top_to_bottom_field */
    int wlog[2], ww[2][16][3], wo[2][16][3];   /* explicit weighted prediction: log2 denominators (luma, chroma), weight / offset [list][ref][Y,Cb,Cr] */
    uint8_t *recon_buf; int recon_frames;
    int max_lt_idx;                         /* MaxLongTermFrameIdx, -1 = "no long-term frame indices" */
    int n_mod[2], mod_idc[2][8], mod_val[2][8];   /* ref_pic_list_modification of the current picture */
    int n_mmco, mmco_op[12], mmco_a[12], mmco_b[12], idr_long;   /* dec_ref_pic_marking of the current picture */
    int16_t tex[256][256];
    int next_id;
    int decoded_mask;
    FILE *recon;
    long stat_bits_mb[8];
    int cabac; CabEnc cab; int last_dqp;   /* entropy_coding_mode_flag; mb_qp_delta of the previous MB != 0 */
} Enc;

static void frame_alloc(Frame *f, int W, int H, int hp) {
    memset(f, 0, sizeof *f);
    f->sy = W + 2 * PAD; f->sc = W / 2 + PAD;
    f->by = (uint8_t *)calloc((size_t)f->sy * (H + 2 * PAD), 1);
    f->bu = (uint8_t *)calloc((size_t)f->sc * (H / 2 + PAD), 1);
    f->bv = (uint8_t *)calloc((size_t)f->sc * (H / 2 + PAD), 1);
    f->y = f->by + PAD * f->sy + PAD; f->u = f->bu + (PAD / 2) * f->sc + PAD / 2; f->v = f->bv + (PAD / 2) * f->sc + PAD / 2;
    if (hp) {
        f->bhb = (uint8_t *)calloc((size_t)f->sy * (H + 2 * PAD), 1); f->bhh = (uint8_t *)calloc((size_t)f->sy * (H + 2 * PAD), 1);
        f->bhj = (uint8_t *)calloc((size_t)f->sy * (H + 2 * PAD), 1);
        f->hb = f->bhb + PAD * f->sy + PAD; f->hh = f->bhh + PAD * f->sy + PAD; f->hj = f->bhj + PAD * f->sy + PAD;
    }
}
static void frame_free(Frame *f) { free(f->by); free(f->bu); free(f->bv); free(f->bhb); free(f->bhh); free(f->bhj); }
static void pad_plane(uint8_t *o, int stride, int w, int h, int pad) {
    for (int y = 0; y < h; y++) { memset(o + y * stride - pad, o[y * stride], pad); memset(o + y * stride + w, o[y * stride + w - 1], pad); }
    for (int y = 1; y <= pad; y++) { memcpy(o - y * stride - pad, o - pad, w + 2 * pad);
        memcpy(o + (h - 1 + y) * stride - pad, o + (h - 1) * stride - pad, w + 2 * pad); }
}
static inline int tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }
/* pad the reconstructed frame and build the three half-sample planes used by motion search */
static void frame_finish_ref(Frame *f, int W, int H) {
    pad_plane(f->y, f->sy, W, H, PAD); pad_plane(f->u, f->sc, W / 2, H / 2, PAD / 2); pad_plane(f->v, f->sc, W / 2, H / 2, PAD / 2);
    int m = PAD - 4, st = f->sy;
    int tw = W + 2 * m, th = H + 2 * m + 5;
    int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)tw * th);     /* unclipped horizontal 6-tap, rows -m-2 .. H+m+2 */
    for (int y = -m - 2; y < H + m + 3; y++) for (int x = -m; x < W + m; x++) {
        const uint8_t *p = f->y + y * st + x;
        tmp[(y + m + 2) * tw + x + m] = (int16_t)tap6(p[-2], p[-1], p[0], p[1], p[2], p[3]);
    }
    for (int y = -m; y < H + m; y++) for (int x = -m; x < W + m; x++) {
        const uint8_t *p = f->y + y * st + x; const int16_t *t = tmp + (y + m + 2) * tw + x + m;
        f->hb[y * st + x] = (uint8_t)CLIP1((t[0] + 16) >> 5);
        f->hh[y * st + x] = (uint8_t)CLIP1((tap6(p[-2 * st], p[-st], p[0], p[st], p[2 * st], p[3 * st]) + 16) >> 5);
        f->hj[y * st + x] = (uint8_t)CLIP1((tap6(t[-2 * tw], t[-tw], t[0], t[tw], t[2 * tw], t[3 * tw]) + 512) >> 10);
    }
    free(tmp);
}

/* ------------------------------ content ------------------------------------ */
static int tri(int p) { p &= 1023; int q = p < 512 ? p : 1023 - p; int x = q - 256; return (x * (512 - ABS(x))) >> 8; /* [-256,256] smooth */ }
static void make_texture(Enc *e) {
    Rng r = { (uint64_t)e->p.seed * 77 + 5 };
    static __thread int16_t n0[64][64];
    for (int y = 0; y < 64; y++) for (int x = 0; x < 64; x++) n0[y][x] = (int16_t)(rnd_n(&r, 65) - 32);
    for (int y = 0; y < 256; y++) for (int x = 0; x < 256; x++) {      /* bilinear-upsampled noise + fine grain */
        int x0 = (x >> 2) & 63, y0 = (y >> 2) & 63, x1 = (x0 + 1) & 63, y1 = (y0 + 1) & 63, fx = x & 3, fy = y & 3;
        int v = (n0[y0][x0] * (4 - fx) * (4 - fy) + n0[y0][x1] * fx * (4 - fy) + n0[y1][x0] * (4 - fx) * fy + n0[y1][x1] * fx * fy) >> 4;
        e->tex[y][x] = (int16_t)(v + rnd_n(&r, 9) - 4);
    }
}
static void render_source(Enc *e, int t) {
    Frame *f = &e->src; int W = e->p.width, H = e->p.height, s = e->p.seed & 0xffff;
    /* background pans at 3/4 px/frame horizontally, 1/4 px/frame vertically (quarter-pel units) */
    int panx4 = 3 * t + (s & 7), pany4 = t + ((s >> 3) & 7);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        int X4 = 4 * x - panx4, Y4 = 4 * y - pany4;
        int v = 128 + ((70 * tri(2 * X4 + Y4 / 2 + s) + 45 * tri(Y4 * 3 - X4 / 2 + 300)) >> 8);
        int tx = (X4 >> 2) & 255, ty = (Y4 >> 2) & 255, fx = X4 & 3, fy = Y4 & 3;
        int t00 = e->tex[ty][tx], t01 = e->tex[ty][(tx + 1) & 255], t10 = e->tex[(ty + 1) & 255][tx], t11 = e->tex[(ty + 1) & 255][(tx + 1) & 255];
        v += (t00 * (4 - fx) * (4 - fy) + t01 * fx * (4 - fy) + t10 * (4 - fx) * fy + t11 * fx * fy) >> 4;
        f->y[y * f->sy + x] = (uint8_t)CLIP1(v);
    }
    /* translating objects with their own texture and velocity */
    for (int k = 0; k < 6; k++) {
        int bw = 32 + 24 * k, bh = 24 + 16 * ((k * 5) % 4);
        int vx4 = (k * 7 + s) % 19 - 9, vy4 = (k * 11 + s) % 13 - 6;
        int ox = (((k * 97 + s * 3) % (W + bw)) * 4 + vx4 * t) >> 2, oy = (((k * 61 + s) % (H + bh)) * 4 + vy4 * t) >> 2;
        ox = ((ox % (W + bw)) + (W + bw)) % (W + bw) - bw; oy = ((oy % (H + bh)) + (H + bh)) % (H + bh) - bh;
        for (int y = MAX(0, oy); y < MIN(H, oy + bh); y++) for (int x = MAX(0, ox); x < MIN(W, ox + bw); x++) {
            int v = 60 + 30 * k + ((40 * tri((x - ox) * 24 + (y - oy) * 9 * (k + 1))) >> 8) + e->tex[(y - oy + 40 * k) & 255][(x - ox + 13 * k) & 255] / 2;
            f->y[y * f->sy + x] = (uint8_t)CLIP1(v);
        }
    }
    for (int y = 0; y < H / 2; y++) for (int x = 0; x < W / 2; x++) {
        int X4 = 8 * x - panx4, Y4 = 8 * y - pany4;
        f->u[y * f->sc + x] = (uint8_t)CLIP1(128 + ((50 * tri(X4 / 2 + Y4 + 100 + s)) >> 8) + e->tex[(Y4 >> 3) & 255][(X4 >> 3) & 255] / 4);
        f->v[y * f->sc + x] = (uint8_t)CLIP1(128 + ((50 * tri(Y4 - X4 / 3 + 700)) >> 8) - e->tex[(X4 >> 3) & 255][(Y4 >> 3) & 255] / 4);
    }
    /* replicate into the coded-size area beyond the display size */
    for (int y = 0; y < e->H; y++) { int yy = MIN(y, H - 1);
        if (yy != y) memcpy(f->y + y * f->sy, f->y + yy * f->sy, W);
        for (int x = W; x < e->W; x++) f->y[y * f->sy + x] = f->y[y * f->sy + W - 1]; }
    for (int y = 0; y < e->H / 2; y++) { int yy = MIN(y, H / 2 - 1);
        if (yy != y) { memcpy(f->u + y * f->sc, f->u + yy * f->sc, W / 2); memcpy(f->v + y * f->sc, f->v + yy * f->sc, W / 2); }
        for (int x = W / 2; x < e->W / 2; x++) { f->u[y * f->sc + x] = f->u[y * f->sc + W / 2 - 1]; f->v[y * f->sc + x] = f->v[y * f->sc + W / 2 - 1]; } }
}

/* ------------------------------ prediction helpers -------------------------- */
static MbE *mb_avail(Enc *e, int mx, int my) {
    if (mx < 0 || my < 0 || mx >= e->mbw || my >= e->mbh) return NULL;
    MbE *m = &e->mbs[my * e->mbw + mx];
    return m->slice == e->slice_id ? m : NULL;
}
static int intra_ok(Enc *e, MbE *m) { return m && (!e->p.cip || m->intra); }

static inline int refpx(const uint8_t *p, int st, int w, int h, int x, int y) { return p[CLIP3(0, h - 1, y) * st + CLIP3(0, w - 1, x)]; }
/* spec-literal luma sample (8.4.2.2.1), coordinate clamping */
static int luma_sample(const Frame *r, int W, int H, int xi, int yi, int fx, int fy) {
#define P(dx, dy) refpx(r->y, r->sy, W, H, xi + (dx), yi + (dy))
#define HB(dy) tap6(P(-2, dy), P(-1, dy), P(0, dy), P(1, dy), P(2, dy), P(3, dy))
#define VH(dx) tap6(P(dx, -2), P(dx, -1), P(dx, 0), P(dx, 1), P(dx, 2), P(dx, 3))
    int G = P(0, 0);
    if (!fx && !fy) return G;
    int b = CLIP1((HB(0) + 16) >> 5), h = CLIP1((VH(0) + 16) >> 5);
    if (!fy) return fx == 2 ? b : (fx == 1 ? (G + b + 1) >> 1 : (P(1, 0) + b + 1) >> 1);
    if (!fx) return fy == 2 ? h : (fy == 1 ? (G + h + 1) >> 1 : (P(0, 1) + h + 1) >> 1);
    int j = CLIP1((tap6(HB(-2), HB(-1), HB(0), HB(1), HB(2), HB(3)) + 512) >> 10);
    int s = CLIP1((HB(1) + 16) >> 5), m = CLIP1((VH(1) + 16) >> 5);
    if (fx == 2 && fy == 2) return j;
    if (fx == 2) return fy == 1 ? (b + j + 1) >> 1 : (j + s + 1) >> 1;
    if (fy == 2) return fx == 1 ? (h + j + 1) >> 1 : (j + m + 1) >> 1;
    if (fx == 1 && fy == 1) return (b + h + 1) >> 1;
    if (fx == 3 && fy == 1) return (b + m + 1) >> 1;
    if (fx == 1 && fy == 3) return (h + s + 1) >> 1;
    return (m + s + 1) >> 1;
#undef P
#undef HB
#undef VH
}
/* fast quarter-sample fetch from the half-pel planes (motion search only) */
static inline int qpel_fast(const Frame *r, int x, int y, int fx, int fy) {
    int st = r->sy; const uint8_t *G = r->y + y * st + x, *b = r->hb + y * st + x, *h = r->hh + y * st + x, *j = r->hj + y * st + x;
    switch (fy * 4 + fx) {
    case 0: return G[0]; case 2: return b[0]; case 8: return h[0]; case 10: return j[0];
    case 1: return (G[0] + b[0] + 1) >> 1; case 3: return (G[1] + b[0] + 1) >> 1;
    case 4: return (G[0] + h[0] + 1) >> 1; case 12: return (G[st] + h[0] + 1) >> 1;
    case 5: return (b[0] + h[0] + 1) >> 1; case 7: return (b[0] + h[1] + 1) >> 1;
    case 13: return (h[0] + b[st] + 1) >> 1; case 15: return (h[1] + b[st] + 1) >> 1;
    case 6: return (b[0] + j[0] + 1) >> 1; case 14: return (j[0] + b[st] + 1) >> 1;
    case 9: return (h[0] + j[0] + 1) >> 1; default: return (j[0] + h[1] + 1) >> 1;
    }
}
static int sad_inter(Enc *e, const Frame *r, int px, int py, int w, int h, int mvx, int mvy) {
    int sad = 0, x0 = px + (mvx >> 2), y0 = py + (mvy >> 2), fx = mvx & 3, fy = mvy & 3;
    const Frame *s = &e->src;
    if (!fx && !fy) { for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) sad += ABS(s->y[(py + y) * s->sy + px + x] - r->y[(y0 + y) * r->sy + x0 + x]); }
    else for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) sad += ABS(s->y[(py + y) * s->sy + px + x] - qpel_fast(r, x0 + x, y0 + y, fx, fy));
    return sad;
}
static int mv_legal(Enc *e, int px, int py, int w, int h, int mvx, int mvy) {
    int lim = PAD - 8;
    int x0 = px + (mvx >> 2), y0 = py + (mvy >> 2);
    return x0 >= -lim && y0 >= -lim && x0 + w <= e->W + lim && y0 + h <= e->H + lim && ABS(mvx) < 4 * 120 && ABS(mvy) < 4 * 120;
}
static void mc_block(Enc *e, const Frame *r, int px, int py, int w, int h, int mvx, int mvy) {
    Frame *c = &e->cur;
    static int check = -1; if (check < 0) check = getenv("H264GEN_CHECK") != NULL;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
        int v = qpel_fast(r, px + x + (mvx >> 2), py + y + (mvy >> 2), mvx & 3, mvy & 3);
        if (check && v != luma_sample(r, e->W, e->H, px + x + (mvx >> 2), py + y + (mvy >> 2), mvx & 3, mvy & 3)) {
            fprintf(stderr, "h264gen: fast/literal MC mismatch\n"); abort(); }
        c->y[(py + y) * c->sy + px + x] = (uint8_t)v;
    }
    /* 8.4.1.4: a field that predicts from a field of the other parity shifts the chroma vector by a quarter chroma sample (Table 8-9): the chroma
       lines of the two parities lie a quarter of a field line apart relative to the luma lines */
    if (e->field) mvy += r->parity == (e->field - 1) ? 0 : (r->parity ? -2 : 2);
    int cw = e->W / 2, ch = e->H / 2, fx = mvx & 7, fy = mvy & 7;
    for (int pl = 0; pl < 2; pl++) {
        const uint8_t *rp = pl ? r->v : r->u; uint8_t *dp = pl ? c->v : c->u;
        for (int y = 0; y < h / 2; y++) for (int x = 0; x < w / 2; x++) {
            int xi = px / 2 + x + (mvx >> 3), yi = py / 2 + y + (mvy >> 3);
            int A = refpx(rp, r->sc, cw, ch, xi, yi), B = refpx(rp, r->sc, cw, ch, xi + 1, yi), C = refpx(rp, r->sc, cw, ch, xi, yi + 1),
                D = refpx(rp, r->sc, cw, ch, xi + 1, yi + 1);
            dp[(py / 2 + y) * c->sc + px / 2 + x] = (uint8_t)(((8 - fx) * (8 - fy) * A + fx * (8 - fy) * B + (8 - fx) * fy * C + fx * fy * D + 32) >> 6);
        }
    }
}

/* neighbour motion data (8.4.1.3.2) */
typedef struct { int avail, ref, mv[2]; } Nbr;
static Nbr nbr_get(Enc *e, int mx, int my, MbE *cur, int bx, int by) {
    Nbr n = {0, -1, {0, 0}}; MbE *m; int rx, ry;
    if (by < 0) { ry = 3;
        if (bx < 0) { m = mb_avail(e, mx - 1, my - 1); rx = 3; }
        else if (bx > 3) { m = mb_avail(e, mx + 1, my - 1); rx = bx - 4; }
        else { m = mb_avail(e, mx, my - 1); rx = bx; } }
    else if (bx < 0) { m = mb_avail(e, mx - 1, my); rx = 3; ry = by; }
    else if (bx > 3) return n;
    else { if (!(e->decoded_mask >> (by * 4 + bx) & 1)) return n; m = cur; rx = bx; ry = by; }
    if (!m) return n;
    n.avail = 1;
    if (m->intra) return n;
    n.ref = m->ref[(ry >> 1) * 2 + (rx >> 1)];
    if (n.ref >= 0) { n.mv[0] = m->mv[ry * 4 + rx][0]; n.mv[1] = m->mv[ry * 4 + rx][1]; }
    return n;
}
static int med3(int a, int b, int c) { return a + b + c - MAX(a, MAX(b, c)) - MIN(a, MIN(b, c)); }
static void pred_mv(Enc *e, int mx, int my, MbE *cur, int bx, int by, int bw, int ref, int shape, int part, int out[2]) {
    Nbr A = nbr_get(e, mx, my, cur, bx - 1, by), B = nbr_get(e, mx, my, cur, bx, by - 1), C = nbr_get(e, mx, my, cur, bx + bw, by - 1);
    if (!C.avail) C = nbr_get(e, mx, my, cur, bx - 1, by - 1);
    if (shape == 1) { if (part == 0 && B.ref == ref) { out[0] = B.mv[0]; out[1] = B.mv[1]; return; } if (part == 1 && A.ref == ref) { out[0] = A.mv[0];
        out[1] = A.mv[1]; return; } }
    if (shape == 2) { if (part == 0 && A.ref == ref) { out[0] = A.mv[0]; out[1] = A.mv[1]; return; } if (part == 1 && C.ref == ref) { out[0] = C.mv[0];
        out[1] = C.mv[1]; return; } }
    if (!B.avail && !C.avail && A.avail) { B = A; C = A; }
    int ma = A.ref == ref, mb = B.ref == ref, mc = C.ref == ref;
    if (ma + mb + mc == 1) { Nbr *n = ma ? &A : (mb ? &B : &C); out[0] = n->mv[0]; out[1] = n->mv[1]; }
    else { out[0] = med3(A.mv[0], B.mv[0], C.mv[0]); out[1] = med3(A.mv[1], B.mv[1], C.mv[1]); }
}
static void skip_mv(Enc *e, int mx, int my, MbE *cur, int out[2]) {
    out[0] = out[1] = 0;
    if (!mb_avail(e, mx - 1, my) || !mb_avail(e, mx, my - 1)) return;
    Nbr A = nbr_get(e, mx, my, cur, -1, 0), B = nbr_get(e, mx, my, cur, 0, -1);
    if ((A.ref == 0 && !A.mv[0] && !A.mv[1]) || (B.ref == 0 && !B.mv[0] && !B.mv[1])) return;
    pred_mv(e, mx, my, cur, 0, 0, 4, 0, 0, 0, out);
}
static void store_mv(Enc *e, MbE *m, int bx, int by, int bw, int bh, int mvx, int mvy) {
    for (int y = by; y < by + bh; y++) for (int x = bx; x < bx + bw; x++) { m->mv[y * 4 + x][0] = (int16_t)mvx; m->mv[y * 4 + x][1] = (int16_t)mvy;
        e->decoded_mask |= 1 << (y * 4 + x); }
}

/* ------------------------------ intra prediction ---------------------------- */
static inline int bX(int blk) { return (blk & 1) + 2 * ((blk >> 2) & 1); }
static inline int bY(int blk) { return ((blk >> 1) & 1) + 2 * (blk >> 3); }
/* which Intra4x4 modes are usable for block blk; fills edge arrays */
static __thread int g_nc_corner;      /* GenParams.nc_corner of the stream being generated */
static __thread int g_i4_avail_d;     /* p[-1,-1] of the block last given to i4_edges() is available (8.3.1.2: modes 4, 5, 6 need it) */
static void i4_edges(Enc *e, int mx, int my, int blk, int *T, int *L, int *aA, int *aB) {
    Frame *c = &e->cur; int bx = bX(blk), by = bY(blk);
    uint8_t *d = c->y + (my * 16 + by * 4) * c->sy + mx * 16 + bx * 4; int st = c->sy;
    int availA = bx > 0 || intra_ok(e, mb_avail(e, mx - 1, my)), availB = by > 0 || intra_ok(e, mb_avail(e, mx, my - 1));
    int availD = (bx > 0 && by > 0) ? 1 : (bx > 0 ? intra_ok(e, mb_avail(e, mx, my - 1)) : (by > 0 ? intra_ok(e, mb_avail(e, mx - 1, my)) : intra_ok(e,
        mb_avail(e, mx - 1, my - 1))));
    int availC;
    if (by == 0) availC = intra_ok(e, bx < 3 ? mb_avail(e, mx, my - 1) : mb_avail(e, mx + 1, my - 1));
    else availC = !(bx == 3 || blk == 3 || blk == 11 || blk == 7 || blk == 13 || blk == 15);
    for (int i = 0; i < 4; i++) { T[i] = availB ? d[-st + i] : 128; L[i] = availA ? d[i * st - 1] : 128; }
    for (int i = 4; i < 8; i++) T[i] = (availB && availC) ? d[-st + i] : T[3];
    T[-1] = L[-1] = availD ? d[-st - 1] : 128;
    *aA = availA; *aB = availB; g_i4_avail_d = availD || g_nc_corner;
}
/* (with constrained_intra_pred the corner can be the only unavailable neighbour; until round 1 this function did not look at it and the
   generator wrote non-conforming Horizontal-Down / Diagonal-Down-Right / Vertical-Right blocks, which is how the decoders' handling of
   unavailable samples came to be compared) */
static int i4_mode_ok(int mode, int aA, int aB) {
    if (mode == 2) return 1;
    if (mode == 0 || mode == 3 || mode == 7) return aB;
    if (mode == 1 || mode == 8) return aA;
    return aA && aB && g_i4_avail_d;
}
static void i4_predict(int mode, const int *T, const int *L, int aA, int aB, int *p) {
    switch (mode) {
    case 0: for (int k = 0; k < 16; k++) p[k] = T[k & 3]; break;
    case 1: for (int k = 0; k < 16; k++) p[k] = L[k >> 2]; break;
    case 2: { int dc = aA && aB ? (T[0] + T[1] + T[2] + T[3] + L[0] + L[1] + L[2] + L[3] + 4) >> 3 : aA ? (L[0] + L[1] + L[2] + L[3] + 2) >> 2 : aB ?
        (T[0] + T[1] + T[2] + T[3] + 2) >> 2 : 128;
        for (int k = 0; k < 16; k++) p[k] = dc; break; }
    case 3: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) p[y * 4 + x] = (x == 3 &&
        y == 3) ? (T[6] + 3 * T[7] + 2) >> 2 : (T[x + y] + 2 * T[x + y + 1] + T[x + y + 2] + 2) >> 2;
    break;
    case 4: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) p[y * 4 + x] = x > y ? (T[x - y - 2] + 2 * T[x - y - 1] + T[x - y] + 2) >> 2 : x < y ?
        (L[y - x - 2] + 2 * L[y - x - 1] + L[y - x] + 2) >> 2 : (T[0] + 2 * T[-1] + L[0] + 2) >> 2;
    break;
    case 5: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { int z = 2 * x - y, i = x - (y >> 1);
            p[y * 4 + x] = z >= 0 ? ((z & 1) ? (T[i - 2] + 2 * T[i - 1] + T[i] + 2) >> 2 : (T[i - 1] + T[i] + 1) >> 1) : z == -1 ?
                (L[0] + 2 * T[-1] + T[0] + 2) >> 2 : (L[y - 1] + 2 * L[y - 2] + L[y - 3] + 2) >> 2; } break;
    case 6: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { int z = 2 * y - x, i = y - (x >> 1);
            p[y * 4 + x] = z >= 0 ? ((z & 1) ? (L[i - 2] + 2 * L[i - 1] + L[i] + 2) >> 2 : (L[i - 1] + L[i] + 1) >> 1) : z == -1 ?
                (L[0] + 2 * T[-1] + T[0] + 2) >> 2 : (T[x - 1] + 2 * T[x - 2] + T[x - 3] + 2) >> 2; } break;
    case 7: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { int i = x + (y >> 1);
        p[y * 4 + x] = (y & 1) ? (T[i] + 2 * T[i + 1] + T[i + 2] + 2) >> 2 : (T[i] + T[i + 1] + 1) >> 1; } break;
    default: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { int z = x + 2 * y, i = y + (x >> 1);
            p[y * 4 + x] = z > 5 ? L[3] : z == 5 ? (L[2] + 3 * L[3] + 2) >> 2 : (z & 1) ? (L[i] + 2 * L[i + 1] + L[i + 2] + 2) >> 2 :
                (L[i] + L[i + 1] + 1) >> 1; } break;
    }
}
/* n x n block prediction for Intra16x16 (n=16, modes V,H,DC,Plane) and chroma (n=8, modes DC,H,V,Plane mapped by caller) */
static void big_predict(const uint8_t *d, int st, int n, int kind /*0 V 1 H 2 DC 3 plane*/, int aA, int aB, int *p) {
    if (kind == 0) { for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) p[y * n + x] = d[-st + x]; }
    else if (kind == 1) { for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) p[y * n + x] = d[y * st - 1]; }
    else if (kind == 3) {
        int Hh = 0, V = 0, h2 = n / 2;
        for (int k = 0; k < h2; k++) { Hh += (k + 1) * (d[-st + h2 + k] - d[-st + h2 - 2 - k]);
            V += (k + 1) * (d[(h2 + k) * st - 1] - d[(h2 - 2 - k) * st - 1]); }
        int a = 16 * (d[(n - 1) * st - 1] + d[-st + n - 1]);
        int b = n == 16 ? (5 * Hh + 32) >> 6 : (34 * Hh + 32) >> 6, c = n == 16 ? (5 * V + 32) >> 6 : (34 * V + 32) >> 6;
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) p[y * n + x] = CLIP1((a + b * (x - (h2 - 1)) + c * (y - (h2 - 1)) + 16) >> 5);
    } else if (n == 16) {
        int s1 = 0, s2 = 0; for (int i = 0; i < 16; i++) { if (aB) s1 += d[-st + i]; if (aA) s2 += d[i * st - 1]; }
        int dc = aA && aB ? (s1 + s2 + 16) >> 5 : aA ? (s2 + 8) >> 4 : aB ? (s1 + 8) >> 4 : 128;
        for (int k = 0; k < 256; k++) p[k] = dc;
    } else {
        for (int by = 0; by < 2; by++) for (int bx = 0; bx < 2; bx++) {
            int s1 = 0, s2 = 0, dc; for (int i = 0; i < 4; i++) { if (aB) s1 += d[-st + bx * 4 + i]; if (aA) s2 += d[(by * 4 + i) * st - 1]; }
            if (bx == by) dc = aA && aB ? (s1 + s2 + 4) >> 3 : aA ? (s2 + 2) >> 2 : aB ? (s1 + 2) >> 2 : 128;
            else if (bx == 1) dc = aB ? (s1 + 2) >> 2 : aA ? (s2 + 2) >> 2 : 128;
            else dc = aA ? (s2 + 2) >> 2 : aB ? (s1 + 2) >> 2 : 128;
            for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) p[(by * 4 + y) * 8 + bx * 4 + x] = dc;
        }
    }
}

/* ------------------------------ transform / quant --------------------------- */
static void fdct4(const int *x, int *w) {
    int t[16];
    for (int i = 0; i < 4; i++) { const int *r = x + 4 * i; int a = r[0] + r[3], b = r[1] + r[2], c = r[1] - r[2], d = r[0] - r[3];
        t[4 * i] = a + b; t[4 * i + 1] = 2 * d + c; t[4 * i + 2] = a - b; t[4 * i + 3] = d - 2 * c; }
    for (int j = 0; j < 4; j++) { int a = t[j] + t[12 + j], b = t[4 + j] + t[8 + j], c = t[4 + j] - t[8 + j], d = t[j] - t[12 + j];
        w[j] = a + b; w[4 + j] = 2 * d + c; w[8 + j] = a - b; w[12 + j] = d - 2 * c; }
}
static inline int pos_class(int k) { int i = k >> 2, j = k & 3; return (!(i & 1) && !(j & 1)) ? 0 : ((i & 1) && (j & 1)) ? 1 : 2; }
static int quant1(int w, int mf, int f, int shift) { int z = (ABS(w) * mf + f) >> shift; z = MIN(z, 2000); return w < 0 ? -z : z; }
static void idct4_add(const int *dq, uint8_t *dst, int st) {
    int f[16];
    for (int i = 0; i < 4; i++) { const int *d = dq + 4 * i; int e0 = d[0] + d[2], e1 = d[0] - d[2], e2 = (d[1] >> 1) - d[3], e3 = d[1] + (d[3] >> 1);
        f[4 * i] = e0 + e3; f[4 * i + 1] = e1 + e2; f[4 * i + 2] = e1 - e2; f[4 * i + 3] = e0 - e3; }
    for (int j = 0; j < 4; j++) { int g0 = f[j] + f[8 + j], g1 = f[j] - f[8 + j], g2 = (f[4 + j] >> 1) - f[12 + j], g3 = f[4 + j] + (f[12 + j] >> 1);
        int r0 = (g0 + g3 + 32) >> 6, r1 = (g1 + g2 + 32) >> 6, r2 = (g1 - g2 + 32) >> 6, r3 = (g0 - g3 + 32) >> 6;
        dst[j] = (uint8_t)CLIP1(dst[j] + r0); dst[st + j] = (uint8_t)CLIP1(dst[st + j] + r1); dst[2 * st + j] = (uint8_t)CLIP1(dst[2 * st + j] + r2);
            dst[3 * st + j] = (uint8_t)CLIP1(dst[3 * st + j] + r3); }
}
/* scaling matrices in effect (8.5.9), raster order: 4x4 lists Intra Y/Cb/Cr, Inter Y/Cb/Cr; 8x8 lists Intra Y, Inter Y */
static __thread int g_w4[6][16], g_w8[2][64], g_wlist;          /* g_wlist: list used by the block being coded */
static inline int dequant_ac(int c, int qp, int k) { int ls = g_w4[g_wlist][k] * norm4[qp % 6][pos_class(k)];
    return qp >= 24 ? (c * ls) << (qp / 6 - 4) : (c * ls + (1 << (3 - qp / 6))) >> (4 - qp / 6); }

/* ------------------------------ CAVLC writer -------------------------------- */
static void put_level_code(BitW *w, int code, int sl) {
    if (sl == 0) {
        if (code < 14) { bw_put(w, code, 0); bw_put(w, 1, 1); }
        else if (code < 30) { bw_put(w, 14, 0); bw_put(w, 1, 1); bw_put(w, 4, code - 14); }
        else { bw_put(w, 15, 0); bw_put(w, 1, 1); bw_put(w, 12, code - 30); }
    } else {
        if (code < (15 << sl)) { bw_put(w, code >> sl, 0); bw_put(w, 1, 1); bw_put(w, sl, code & ((1 << sl) - 1)); }
        else { bw_put(w, 15, 0); bw_put(w, 1, 1); bw_put(w, 12, code - (15 << sl)); }
    }
}
/* coef: scan-order coefficients; returns total_coeff */
static int write_block(BitW *w, const int *coef, int max_num, int nC) {
    int lev[16], pos[16], total = 0;
    for (int i = max_num - 1; i >= 0; i--) if (coef[i]) { lev[total] = coef[i]; pos[total] = i; total++; }
    int t1 = 0;
    while (t1 < total && t1 < 3 && ABS(lev[t1]) == 1) t1++;
    int idx = 4 * total + t1;
    if (nC == -1) bw_put(w, cdc_len[idx], cdc_bits[idx]);
    else { int t = nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3; bw_put(w, ct_len[t][idx], ct_bits[t][idx]); }
    if (!total) return 0;
    int sl = (total > 10 && t1 < 3) ? 1 : 0;
    for (int i = 0; i < total; i++) {
        if (i < t1) { bw_put(w, 1, lev[i] < 0); continue; }
        int code = lev[i] > 0 ? 2 * lev[i] - 2 : -2 * lev[i] - 1;
        if (i == t1 && t1 < 3) code -= 2;
        put_level_code(w, code, sl);
        if (sl == 0) sl = 1;
        if (ABS(lev[i]) > (3 << (sl - 1)) && sl < 6) sl++;
    }
    int zeros = pos[0] + 1 - total;
    if (total < max_num) {
        if (max_num == 4) bw_put(w, ctz_len[total - 1][zeros], ctz_bits[total - 1][zeros]);
        else bw_put(w, tz_len[total - 1][zeros], tz_bits[total - 1][zeros]);
    }
    int left = zeros;
    for (int i = 0; i < total - 1 && left > 0; i++) {
        int run = pos[i] - pos[i + 1] - 1, t = MIN(left, 7) - 1;
        bw_put(w, rb_len[t][run], rb_bits[t][run]);
        left -= run;
    }
    return total;
}

/* ------------------------------ MB coding state ------------------------------ */
typedef struct {
    int type;                 /* 0 P16x16 1 P16x8 2 P8x16 3 P8x8 4 skip 5 I4x4 6 I16x16 7 PCM */
    int sub[4];
    int refs[4];
    int i16mode, cmode;
    int i4modes[16];
    int cbp;
    int luma[16][16];         /* quantised levels per raster 4x4 block, raster coefficient order */
    int dc16[16];
    int cdc[2][4], cac[2][4][16];
    int dqp;
    int t8; int luma8[4][64]; /* transform_size_8x8_flag; levels per 8x8 block, raster */
} MbCode;

static int nC_luma(Enc *e, int mx, int my, MbE *cur, int bx, int by) {
    int aA = 0, aB = 0, nA = 0, nB = 0; MbE *m;
    if (bx > 0) { aA = 1; nA = cur->tc[by * 4 + bx - 1]; } else if ((m = mb_avail(e, mx - 1, my))) { aA = 1; nA = m->tc[by * 4 + 3]; }
    if (by > 0) { aB = 1; nB = cur->tc[(by - 1) * 4 + bx]; } else if ((m = mb_avail(e, mx, my - 1))) { aB = 1; nB = m->tc[12 + bx]; }
    return aA && aB ? (nA + nB + 1) >> 1 : aA ? nA : aB ? nB : 0;
}
static int nC_chroma(Enc *e, int mx, int my, MbE *cur, int pl, int bx, int by) {
    int o = 16 + 4 * pl, aA = 0, aB = 0, nA = 0, nB = 0; MbE *m;
    if (bx > 0) { aA = 1; nA = cur->tc[o + by * 2]; } else if ((m = mb_avail(e, mx - 1, my))) { aA = 1; nA = m->tc[o + by * 2 + 1]; }
    if (by > 0) { aB = 1; nB = cur->tc[o + bx]; } else if ((m = mb_avail(e, mx, my - 1))) { aB = 1; nB = m->tc[o + 2 + bx]; }
    return aA && aB ? (nA + nB + 1) >> 1 : aA ? nA : aB ? nB : 0;
}
static int chroma_qp_of(Enc *e, int qp) { int q = CLIP3(0, 51, qp + e->p.chroma_qp_off); return q < 30 ? q : qpc_tab[q - 30]; }

/* transform+quantise+reconstruct one luma 4x4 block (pred already in cur frame). returns nonzero count */
static int code_luma4(Enc *e, int px, int py, int qp, int intra, int *levels /*raster*/) {
    Frame *c = &e->cur, *s = &e->src; int x[16], w[16], nz = 0;
    for (int k = 0; k < 16; k++) x[k] = s->y[(py + (k >> 2)) * s->sy + px + (k & 3)] - c->y[(py + (k >> 2)) * c->sy + px + (k & 3)];
    fdct4(x, w);
    int shift = 15 + qp / 6, f = (1 << shift) / (intra ? 3 : 6);
    g_wlist = intra ? 0 : 3;
    for (int k = 0; k < 16; k++) { levels[k] = quant1(w[k], quant_mf[qp % 6][pos_class(k)] * 16 / g_w4[g_wlist][k], f, shift); nz += levels[k] != 0; }
    if (nz) { int dq[16]; for (int k = 0; k < 16; k++) dq[k] = dequant_ac(levels[k], qp, k); idct4_add(dq, c->y + py * c->sy + px, c->sy); }
    return nz;
}

/* chroma residual for one plane; pred in cur. fills cdc/cac; returns flags: bit0 dc nonzero, bit1 ac nonzero */
static int code_chroma(Enc *e, int mx, int my, int pl, int qpc, int intra, MbCode *mc) {
    Frame *c = &e->cur, *s = &e->src; uint8_t *cp = (pl ? c->v : c->u) + my * 8 * c->sc + mx * 8;
    const uint8_t *sp = (pl ? s->v : s->u) + my * 8 * s->sc + mx * 8;
    int w[4][16], dcs[4], flags = 0;
    int shift = 15 + qpc / 6, f = (1 << shift) / (intra ? 3 : 6);
    for (int k = 0; k < 4; k++) {
        int x[16]; int ox = (k & 1) * 4, oy = (k >> 1) * 4;
        for (int i = 0; i < 16; i++) x[i] = sp[(oy + (i >> 2)) * s->sc + ox + (i & 3)] - cp[(oy + (i >> 2)) * c->sc + ox + (i & 3)];
        fdct4(x, w[k]); dcs[k] = w[k][0];
        mc->cac[pl][k][0] = 0;
        for (int i = 1; i < 16; i++) { mc->cac[pl][k][i] = quant1(w[k][i], quant_mf[qpc % 6][pos_class(i)] * 16 / g_w4[(intra ? 0 : 3) + 1 + pl][i], f, shift);
            if (mc->cac[pl][k][i]) flags |= 2; }
    }
    int h[4] = { dcs[0] + dcs[1] + dcs[2] + dcs[3], dcs[0] - dcs[1] + dcs[2] - dcs[3], dcs[0] + dcs[1] - dcs[2] - dcs[3], dcs[0] - dcs[1] - dcs[2] + dcs[3] };
    for (int k = 0; k < 4; k++) { mc->cdc[pl][k] = quant1(h[k], quant_mf[qpc % 6][0] * 16 / g_w4[(intra ? 0 : 3) + 1 + pl][0], 2 * f, shift + 1);
        if (mc->cdc[pl][k]) flags |= 1; }
    return flags;
}
static void recon_chroma(Enc *e, int mx, int my, int pl, int qpc, const MbCode *mc, int use_dc, int use_ac) {
    Frame *c = &e->cur; uint8_t *cp = (pl ? c->v : c->u) + my * 8 * c->sc + mx * 8;
    if (!use_dc && !use_ac) return;
    g_wlist = (e->mbs[my * e->mbw + mx].intra ? 0 : 3) + 1 + pl;
    const int *d = mc->cdc[pl];
    int f[4] = { d[0] + d[1] + d[2] + d[3], d[0] - d[1] + d[2] - d[3], d[0] + d[1] - d[2] - d[3], d[0] - d[1] - d[2] + d[3] };
    for (int k = 0; k < 4; k++) {
        int dq[16];
        for (int i = 1; i < 16; i++) dq[i] = use_ac ? dequant_ac(mc->cac[pl][k][i], qpc, i) : 0;
        dq[0] = use_dc ? ((f[k] * g_w4[g_wlist][0] * norm4[qpc % 6][0]) << (qpc / 6)) >> 5 : 0;
        idct4_add(dq, cp + (k >> 1) * 4 * c->sc + (k & 1) * 4, c->sc);
    }
}

/* ------------------------------ deblocking (own implementation) --------------- */
/* 8.7.2.1: vectors differ by a whole luma FRAME sample or more; in a field picture a vertical difference of 4 quarter frame samples is 2 quarter field
   samples */
static __thread int g_mvy_limit = 4;
static int mv_far(const int16_t *a, const int16_t *b) { return ABS(a[0] - b[0]) >= 4 || ABS(a[1] - b[1]) >= g_mvy_limit; }
/* mbedge: 0 inner edge, 1 macroblock edge, 2 horizontal macroblock edge of a field picture (8.7.2.1: bS 4 needs frame macroblocks or a vertical edge) */
static int edge_bs(const MbE *p, int bp, const MbE *q, int bq, int mbedge) {
    if (p->intra || q->intra) return mbedge == 1 ? 4 : 3;
    if (((p->nzmask >> bp) & 1) || ((q->nzmask >> bq) & 1)) return 2;
    int pq = (bp >> 3) * 2 + ((bp & 3) >> 1), qq = (bq >> 3) * 2 + ((bq & 3) >> 1);
    /* reference pictures actually used by each side (as a set of at most two), with their vectors */
    int pr[2], qr[2], np = 0, nq = 0; const int16_t *pv[2], *qv[2];
    if (p->ref[pq] >= 0) { pr[np] = p->refid[pq]; pv[np++] = p->mv[bp]; }
    if (p->ref1[pq] >= 0) { pr[np] = p->refid1[pq]; pv[np++] = p->mv1[bp]; }
    if (q->ref[qq] >= 0) { qr[nq] = q->refid[qq]; qv[nq++] = q->mv[bq]; }
    if (q->ref1[qq] >= 0) { qr[nq] = q->refid1[qq]; qv[nq++] = q->mv1[bq]; }
    if (np != nq) return 1;
    if (np == 1) return pr[0] != qr[0] || mv_far(pv[0], qv[0]);
    int straight = pr[0] == qr[0] && pr[1] == qr[1], crossed = pr[0] == qr[1] && pr[1] == qr[0];
    if (!straight && !crossed) return 1;
    if (pr[0] != pr[1]) return straight ? (mv_far(pv[0], qv[0]) || mv_far(pv[1], qv[1])) : (mv_far(pv[0], qv[1]) || mv_far(pv[1], qv[0]));
    return (mv_far(pv[0], qv[0]) || mv_far(pv[1], qv[1])) && (mv_far(pv[0], qv[1]) || mv_far(pv[1], qv[0]));
}
static void db_luma(uint8_t *q, int s, int bS, int a, int b, int ia) {
    int p0 = q[-s], p1 = q[-2 * s], p2 = q[-3 * s], q0 = q[0], q1 = q[s], q2 = q[2 * s];
    if (ABS(p0 - q0) >= a || ABS(p1 - p0) >= b || ABS(q1 - q0) >= b) return;
    int ap = ABS(p2 - p0) < b, aq = ABS(q2 - q0) < b;
    if (bS == 4) {
        int small = ABS(p0 - q0) < (a >> 2) + 2;
        if (ap && small) { int p3 = q[-4 * s]; q[-s] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            q[-2 * s] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2); q[-3 * s] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3); }
        else q[-s] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        if (aq && small) { int q3 = q[3 * s]; q[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3); q[s] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
            q[2 * s] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3); }
        else q[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    } else {
        int t0 = tc0_tab[ia][bS - 1], tc = t0 + ap + aq;
        int dl = CLIP3(-tc, tc, (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3);
        q[-s] = (uint8_t)CLIP1(p0 + dl); q[0] = (uint8_t)CLIP1(q0 - dl);
        if (ap) q[-2 * s] = (uint8_t)(p1 + CLIP3(-t0, t0, (p2 + ((p0 + q0 + 1) >> 1) - 2 * p1) >> 1));
        if (aq) q[s] = (uint8_t)(q1 + CLIP3(-t0, t0, (q2 + ((p0 + q0 + 1) >> 1) - 2 * q1) >> 1));
    }
}
static void db_chroma(uint8_t *q, int s, int bS, int a, int b, int ia) {
    int p0 = q[-s], p1 = q[-2 * s], q0 = q[0], q1 = q[s];
    if (ABS(p0 - q0) >= a || ABS(p1 - p0) >= b || ABS(q1 - q0) >= b) return;
    if (bS == 4) { q[-s] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2); q[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2); }
    else { int tc = tc0_tab[ia][bS - 1] + 1, dl = CLIP3(-tc, tc, (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3); q[-s] = (uint8_t)CLIP1(p0 + dl);
        q[0] = (uint8_t)CLIP1(q0 - dl); }
}
static void deblock_frame(Enc *e) {
    Frame *c = &e->cur;
    g_mvy_limit = e->field ? 2 : 4;
    for (int my = 0; my < e->mbh; my++) for (int mx = 0; mx < e->mbw; mx++) {
        MbE *q = &e->mbs[my * e->mbw + mx];
        if (q->dis_db == 1) continue;
        for (int dir = 0; dir < 2; dir++) for (int ed = 0; ed < 4; ed++) {
            const MbE *p = q;
            if ((ed & 1) && q->t8) continue;                /* 8x8 transform: only 8x8 block edges are filtered */
            if (ed == 0) { int nx = mx - !dir, ny = my - dir; if (nx < 0 || ny < 0) continue; p = &e->mbs[ny * e->mbw + nx];
                if (q->dis_db == 2 && p->slice != q->slice) continue; }
            int bs[4], any = 0;
            for (int k = 0; k < 4; k++) { int bq = dir ? ed * 4 + k : k * 4 + ed; int bp = ed ? (dir ? bq - 4 : bq - 1) : (dir ? 12 + k : k * 4 + 3);
                bs[k] = edge_bs(p, bp, q, bq, ed == 0 ? (dir && e->field ? 2 : 1) : 0); any |= bs[k]; }
            if (!any) continue;
            int qa = (p->qp + q->qp + 1) >> 1, ia = CLIP3(0, 51, qa + q->a_off), ib = CLIP3(0, 51, qa + q->b_off);
            for (int i = 0; i < 16; i++) if (bs[i >> 2]) {
                uint8_t *px = dir ? c->y + (my * 16 + ed * 4) * c->sy + mx * 16 + i : c->y + (my * 16 + i) * c->sy + mx * 16 + ed * 4;
                db_luma(px, dir ? c->sy : 1, bs[i >> 2], alpha_tab[ia], beta_tab[ib], ia);
            }
            if (ed & 1) continue;
            for (int pl = 0; pl < 2; pl++) {
                uint8_t *base = pl ? c->v : c->u;
                int qc = (p->qpc[pl] + q->qpc[pl] + 1) >> 1, ca = CLIP3(0, 51, qc + q->a_off), cb = CLIP3(0, 51, qc + q->b_off);
                for (int i = 0; i < 8; i++) if (bs[i >> 1]) {
                    uint8_t *px = dir ? base + (my * 8 + ed * 2) * c->sc + mx * 8 + i : base + (my * 8 + i) * c->sc + mx * 8 + ed * 2;
                    db_chroma(px, dir ? c->sc : 1, bs[i >> 1], alpha_tab[ca], beta_tab[cb], ca);
                }
            }
        }
    }
}

/* ------------------------------ macroblock encode ---------------------------- */
static int sad16_pred(Enc *e, int mx, int my, const int *p) {
    const Frame *s = &e->src; int sad = 0;
    for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) sad += ABS(s->y[(my * 16 + y) * s->sy + mx * 16 + x] - p[y * 16 + x]);
    return sad;
}

/* ------------------------------ 8x8 transform (own implementation) ---------- */
static const uint8_t zz8[64] = { 0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,
    35,42,49,56,57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63 };
/* 8x8 field scan (Table 8-3, raster index = 8 * y + x) */
static const uint8_t fs8[64] = { 0,8,16,1,9,24,32,17, 2,25,40,48,56,33,10,3, 18,41,49,57,26,11,4,19, 34,42,50,58,27,12,5,20,
    35,43,51,59,28,13,6,21, 36,44,52,60,29,14,22,37, 45,53,61,30,7,15,38,46, 54,62,23,31,39,47,55,63 };
static const int norm8[6][6] = { {20,18,32,19,25,24},{22,19,35,21,28,26},{26,23,42,24,33,31},{28,25,45,26,35,33},{32,28,51,30,40,38},{36,32,58,34,46,43} };
static int cls8(int i, int j) {
    int a = (i & 1) ? 1 : ((i & 3) ? 2 : 0), b = (j & 1) ? 1 : ((j & 3) ? 2 : 0);
    if (a == b) return a;
    if (a + b == 1) return 3;
    return a + b == 2 ? 4 : 5;
}
/* 8.5.13 written with the standard's intermediate names a[], b[] */
static void inv8_1d(const int *in, int *out) {
    int a[8], b[8];
    a[0] = in[0] + in[4]; a[1] = -in[3] + in[5] - in[7] - (in[7] >> 1); a[2] = in[0] - in[4]; a[3] = in[1] + in[7] - in[3] - (in[3] >> 1);
    a[4] = (in[2] >> 1) - in[6]; a[5] = -in[1] + in[7] + in[5] + (in[5] >> 1); a[6] = in[2] + (in[6] >> 1); a[7] = in[3] + in[5] + in[1] + (in[1] >> 1);
    b[0] = a[0] + a[6]; b[1] = a[1] + (a[7] >> 2); b[2] = a[2] + a[4]; b[3] = a[3] + (a[5] >> 2);
    b[4] = a[2] - a[4]; b[5] = (a[3] >> 2) - a[5]; b[6] = a[0] - a[6]; b[7] = a[7] - (a[1] >> 2);
    out[0] = b[0] + b[7]; out[1] = b[2] + b[5]; out[2] = b[4] + b[3]; out[3] = b[6] + b[1];
    out[4] = b[6] - b[1]; out[5] = b[4] - b[3]; out[6] = b[2] - b[5]; out[7] = b[0] - b[7];
}
static void recon8(const int *lev /*raster*/, int qp, int list, uint8_t *dst, int st) {
    int d[64], g[64], col[8], o[8];
    for (int k = 0; k < 64; k++) { int ls = g_w8[list][k] * norm8[qp % 6][cls8(k >> 3, k & 7)];
        d[k] = qp >= 36 ? (lev[k] * ls) << (qp / 6 - 6) : (lev[k] * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6); }
    for (int i = 0; i < 8; i++) inv8_1d(d + 8 * i, g + 8 * i);
    for (int j = 0; j < 8; j++) {
        for (int i = 0; i < 8; i++) col[i] = g[8 * i + j];
        inv8_1d(col, o);
        for (int i = 0; i < 8; i++) dst[i * st + j] = (uint8_t)CLIP1(dst[i * st + j] + ((o[i] + 32) >> 6));
    }
}
/* forward side: projection on the decoder's (orthogonal) reconstruction basis, computed in floating point per QP */
static __thread int g_w8_version;   /* per thread: bench.py generates streams on several threads */
static float *basis8(int qp, int list) {
    static __thread float *tab[2][52]; static __thread int version[2][52];
    if (tab[list][qp] && version[list][qp] == g_w8_version) return tab[list][qp];
    free(tab[list][qp]);
    float *B = (float *)malloc(sizeof(float) * 64 * 65);
    for (int k = 0; k < 64; k++) {
        double d[64], g[64];
        for (int i = 0; i < 64; i++) d[i] = 0;
        d[k] = (double)g_w8[list][k] * norm8[qp % 6][cls8(k >> 3, k & 7)] * (double)(1 << (qp / 6)) / 64.0;
        for (int pass = 0; pass < 2; pass++) {
            for (int r = 0; r < 8; r++) {
                double in[8], a[8], b[8], *src = pass ? g : d;
                for (int i = 0; i < 8; i++) in[i] = pass ? src[8 * i + r] : src[8 * r + i];
                a[0] = in[0] + in[4]; a[1] = -in[3] + in[5] - in[7] - in[7] / 2; a[2] = in[0] - in[4]; a[3] = in[1] + in[7] - in[3] - in[3] / 2;
                a[4] = in[2] / 2 - in[6]; a[5] = -in[1] + in[7] + in[5] + in[5] / 2; a[6] = in[2] + in[6] / 2; a[7] = in[3] + in[5] + in[1] + in[1] / 2;
                b[0] = a[0] + a[6]; b[1] = a[1] + a[7] / 4; b[2] = a[2] + a[4]; b[3] = a[3] + a[5] / 4; b[4] = a[2] - a[4]; b[5] = a[3] / 4 - a[5];
                b[6] = a[0] - a[6]; b[7] = a[7] - a[1] / 4;
                double o[8] = { b[0] + b[7], b[2] + b[5], b[4] + b[3], b[6] + b[1], b[6] - b[1], b[4] - b[3], b[2] - b[5], b[0] - b[7] };
                for (int i = 0; i < 8; i++) { if (pass) d[8 * i + r] = o[i]; else g[8 * r + i] = o[i]; }
            }
        }
        double nn = 0;
        for (int i = 0; i < 64; i++) { B[k * 65 + i] = (float)(d[i] / 64.0); nn += (d[i] / 64.0) * (d[i] / 64.0); }
        B[k * 65 + 64] = (float)nn;
    }
    version[list][qp] = g_w8_version;
    return tab[list][qp] = B;
}
/* transform+quantise+reconstruct one luma 8x8 block (pred already in cur frame); returns nonzero count */
static int code_luma8(Enc *e, int px, int py, int qp, int intra, int *levels /*raster*/) {
    Frame *c = &e->cur, *s = &e->src; float x[64]; int nz = 0;
    const float *B = basis8(qp, intra ? 0 : 1);
    for (int k = 0; k < 64; k++) x[k] = (float)(s->y[(py + (k >> 3)) * s->sy + px + (k & 7)] - c->y[(py + (k >> 3)) * c->sy + px + (k & 7)]);
    float dz = intra ? 0.33f : 0.17f;
    for (int k = 0; k < 64; k++) {
        float acc = 0; const float *b = B + k * 65;
        for (int i = 0; i < 64; i++) acc += x[i] * b[i];
        float v = acc / b[64]; int z = (int)((v < 0 ? -v : v) + dz); z = MIN(z, 2000);
        levels[k] = v < 0 ? -z : z; nz += z != 0;
    }
    if (nz) recon8(levels, qp, intra ? 0 : 1, c->y + py * c->sy + px, c->sy);
    return nz;
}
/* Intra8x8 (8.3.2): raw + filtered reference samples of 8x8 block b8; T/L index -1 = corner */
static void i8_edges(Enc *e, int mx, int my, int b8, int *T /*[-1..15]*/, int *L /*[-1..7]*/, int *aA, int *aB, int *aD) {
    Frame *c = &e->cur; int bx = b8 & 1, by = b8 >> 1;
    uint8_t *d = c->y + (my * 16 + by * 8) * c->sy + mx * 16 + bx * 8; int st = c->sy;
    int availA = bx || intra_ok(e, mb_avail(e, mx - 1, my)), availB = by || intra_ok(e, mb_avail(e, mx, my - 1));
    int availD = (bx && by) ? 1 : (bx ? intra_ok(e, mb_avail(e, mx, my - 1)) : (by ? intra_ok(e, mb_avail(e, mx - 1, my)) : intra_ok(e, mb_avail(e, mx - 1,
        my - 1))));
    int availC = b8 == 0 ? intra_ok(e, mb_avail(e, mx, my - 1)) : (b8 == 1 ? intra_ok(e, mb_avail(e, mx + 1, my - 1)) : b8 == 2);
    int rt[17], rl[9], *t = rt + 1, *l = rl + 1;
    for (int i = 0; i < 16; i++) t[i] = availB ? d[-st + ((i >= 8 && !availC) ? 7 : i)] : 128;
    for (int i = 0; i < 8; i++) l[i] = availA ? d[i * st - 1] : 128;
    t[-1] = l[-1] = availD ? d[-st - 1] : 128;
    for (int i = 0; i < 16; i++) {
        int lo = i == 0 ? (availD ? t[-1] : t[0]) : t[i - 1], hi = i == 15 ? t[15] : t[i + 1];
        T[i] = (lo + 2 * t[i] + hi + 2) >> 2;
    }
    for (int i = 0; i < 8; i++) {
        int lo = i == 0 ? (availD ? l[-1] : l[0]) : l[i - 1], hi = i == 7 ? l[7] : l[i + 1];
        L[i] = (lo + 2 * l[i] + hi + 2) >> 2;
    }
    if (!availD) T[-1] = L[-1] = 128;
    else if (availA && availB) T[-1] = L[-1] = (t[0] + 2 * t[-1] + l[0] + 2) >> 2;
    else if (availB) T[-1] = L[-1] = (3 * t[-1] + t[0] + 2) >> 2;
    else if (availA) T[-1] = L[-1] = (3 * t[-1] + l[0] + 2) >> 2;
    else T[-1] = L[-1] = t[-1];
    *aA = availA; *aB = availB; *aD = availD;
}
static int i8_mode_ok(int mode, int aA, int aB, int aD) {
    if (mode == 2) return 1;
    if (mode == 0 || mode == 3 || mode == 7) return aB;
    if (mode == 1 || mode == 8) return aA;
    return aA && aB && aD;
}
static void i8_predict(int mode, const int *T, const int *L, int aA, int aB, int *p) {
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
        int v;
        if (mode == 0) v = T[x];
        else if (mode == 1) v = L[y];
        else if (mode == 2) { int s1 = 0, s2 = 0; for (int i = 0; i < 8; i++) { s1 += T[i]; s2 += L[i]; }
            v = aA && aB ? (s1 + s2 + 8) >> 4 : aA ? (s2 + 4) >> 3 : aB ? (s1 + 4) >> 3 : 128; }
        else if (mode == 3) v = (x == 7 && y == 7) ? (T[14] + 3 * T[15] + 2) >> 2 : (T[x + y] + 2 * T[x + y + 1] + T[x + y + 2] + 2) >> 2;
        else if (mode == 4) v = x > y ? (T[x - y - 2] + 2 * T[x - y - 1] + T[x - y] + 2) >> 2 : x < y ? (L[y - x - 2] + 2 * L[y - x - 1] + L[y - x] + 2) >> 2 :
            (T[0] + 2 * T[-1] + L[0] + 2) >> 2;
        else if (mode == 5) { int z = 2 * x - y, i = x - (y >> 1);
            v = z >= 0 ? ((z & 1) ? (T[i - 2] + 2 * T[i - 1] + T[i] + 2) >> 2 : (T[i - 1] + T[i] + 1) >> 1) : z == -1 ? (L[0] + 2 * T[-1] + T[0] + 2) >> 2 :
                (L[y - 2 * x - 1] + 2 * L[y - 2 * x - 2] + L[y - 2 * x - 3] + 2) >> 2; }
        else if (mode == 6) { int z = 2 * y - x, i = y - (x >> 1);
            v = z >= 0 ? ((z & 1) ? (L[i - 2] + 2 * L[i - 1] + L[i] + 2) >> 2 : (L[i - 1] + L[i] + 1) >> 1) : z == -1 ? (L[0] + 2 * T[-1] + T[0] + 2) >> 2 :
                (T[x - 2 * y - 1] + 2 * T[x - 2 * y - 2] + T[x - 2 * y - 3] + 2) >> 2; }
        else if (mode == 7) { int i = x + (y >> 1); v = (y & 1) ? (T[i] + 2 * T[i + 1] + T[i + 2] + 2) >> 2 : (T[i] + T[i + 1] + 1) >> 1; }
        else { int z = x + 2 * y, i = y + (x >> 1);
            v = z > 13 ? L[7] : z == 13 ? (L[6] + 3 * L[7] + 2) >> 2 : (z & 1) ? (L[i] + 2 * L[i + 1] + L[i + 2] + 2) >> 2 : (L[i] + L[i + 1] + 1) >> 1; }
        p[y * 8 + x] = v;
    }
}

/* ------------------------------ syntax elements: CAVLC or CABAC --------------- */
static int mb_inxn(const MbE *m) { return m->intra && !m->i16 && !m->pcm; }
static MbE *nb4(Enc *e, int mx, int my, MbE *cur, int bx, int by, int left, int *r) {
    if (left) { if (bx > 0) { *r = by * 4 + bx - 1; return cur; } *r = by * 4 + 3; return mb_avail(e, mx - 1, my); }
    if (by > 0) { *r = (by - 1) * 4 + bx; return cur; }
    *r = 12 + bx; return mb_avail(e, mx, my - 1);
}
static void se_skip_flag(Enc *e, int mx, int my, int skip) {
    MbE *a = mb_avail(e, mx - 1, my), *b = mb_avail(e, mx, my - 1);
    cab_enc(&e->cab, 11 + (a && !a->skip) + (b && !b->skip), skip);
}
static void se_skip_flag_b(Enc *e, int mx, int my, int skip);
static void se_mb_type_b_intra_prefix(Enc *e, int mx, int my);
/* called before every non-skipped macroblock */
static void se_begin_mb(Enc *e, int mx, int my, int *skip_run) {
    if (e->cabac) { if (e->slice_type == 0) se_skip_flag(e, mx, my, 0); else if (e->slice_type == 1) se_skip_flag_b(e, mx, my, 0); }
    else if (*skip_run >= 0) { bw_ue(&e->bw, *skip_run); *skip_run = 0; }
}
/* itype: 0 I_NxN, 1..24 I_16x16 (Table 7-11), 25 I_PCM */
static void se_mb_type_intra(Enc *e, int mx, int my, int itype) {
    int off = e->slice_type == 0 ? 5 : (e->slice_type == 1 ? 23 : 0);
    if (!e->cabac) { bw_ue(&e->bw, itype + off); return; }
    CabEnc *c = &e->cab; int base, in_i = e->slice_type == 2;
    if (in_i) {
        MbE *a = mb_avail(e, mx - 1, my), *b = mb_avail(e, mx, my - 1);
        cab_enc(c, 3 + (a && !mb_inxn(a)) + (b && !mb_inxn(b)), itype != 0);
        base = 5;
    } else if (e->slice_type == 0) { cab_enc(c, 14, 1); cab_enc(c, 17, itype != 0); base = 17; }
    else { se_mb_type_b_intra_prefix(e, mx, my); cab_enc(c, 32, itype != 0); base = 32; }
    if (itype == 0) return;
    cab_term(c, itype == 25);
    if (itype == 25) return;
    int k = itype - 1, cbp_l = k >= 12, cc = (k / 4) % 3, md = k & 3;
    cab_enc(c, base + 1, cbp_l);
    cab_enc(c, base + 2, cc != 0);
    if (cc) cab_enc(c, base + 2 + in_i, cc == 2);
    cab_enc(c, base + 3 + in_i, md >> 1);
    cab_enc(c, base + 3 + 2 * in_i, md & 1);
}
static void se_mb_type_p(Enc *e, int type, int p8ref0) {
    if (!e->cabac) { bw_ue(&e->bw, type == 3 && p8ref0 ? 4 : type); return; }
    CabEnc *c = &e->cab;
    cab_enc(c, 14, 0);
    if (type == 0 || type == 3) { cab_enc(c, 15, 0); cab_enc(c, 16, type == 3); }
    else { cab_enc(c, 15, 1); cab_enc(c, 17, type == 1); }
}
static void se_sub_mb_type(Enc *e, int st) {
    if (!e->cabac) { bw_ue(&e->bw, st); return; }
    CabEnc *c = &e->cab;
    if (st == 0) { cab_enc(c, 21, 1); return; }
    cab_enc(c, 21, 0);
    if (st == 1) { cab_enc(c, 22, 0); return; }
    cab_enc(c, 22, 1); cab_enc(c, 23, st == 2);
}
static void se_ref_idx(Enc *e, int mx, int my, MbE *m, int bx, int by, int nref, int v) {
    if (!e->cabac) { bw_te(&e->bw, nref - 1, v); return; }
    int inc = 0;
    for (int k = 0; k < 2; k++) { int r; MbE *n = nb4(e, mx, my, m, bx, by, k == 0, &r);
        if (n && !n->intra && n->ref[(r >> 3) * 2 + ((r & 3) >> 1)] > 0) inc += k == 0 ? 1 : 2; }
    int ctx = 54 + inc;
    for (int i = 0; i < v; i++) { cab_enc(&e->cab, ctx, 1); ctx = 54 + (i == 0 ? 4 : 5); }
    cab_enc(&e->cab, ctx, 0);
}
static void se_mvd(Enc *e, int mx, int my, MbE *m, int bx, int by, int bw, int bh, int dx, int dy) {
    if (!e->cabac) { bw_se(&e->bw, dx); bw_se(&e->bw, dy); return; }
    CabEnc *c = &e->cab;
    for (int comp = 0; comp < 2; comp++) {
        int d = comp ? dy : dx, a = ABS(d), sum = 0, base = comp ? 47 : 40;
        for (int k = 0; k < 2; k++) { int r; MbE *n = nb4(e, mx, my, m, bx, by, k == 0, &r); if (n && !n->intra) sum += n->mvd[r][comp]; }
        cab_enc(c, base + (sum < 3 ? 0 : sum > 32 ? 2 : 1), a > 0);
        if (!a) continue;
        int v = 1, ctx = 3;
        while (v < 9) { int bin = a > v; cab_enc(c, base + ctx, bin); if (!bin) break; v++; if (ctx < 6) ctx++; }
        if (a >= 9) cab_ueg(c, a - 9, 3);
        cab_byp(c, d < 0);
    }
    for (int y = by; y < by + bh; y++) for (int x = bx; x < bx + bw; x++) { m->mvd[y * 4 + x][0] = (uint8_t)MIN(ABS(dx), 255);
        m->mvd[y * 4 + x][1] = (uint8_t)MIN(ABS(dy), 255); }
}
static void se_t8_flag(Enc *e, int mx, int my, int v) {
    if (!e->cabac) { bw_put(&e->bw, 1, (uint32_t)v); return; }
    MbE *a = mb_avail(e, mx - 1, my), *b = mb_avail(e, mx, my - 1);
    cab_enc(&e->cab, 399 + (a && a->t8) + (b && b->t8), v);
}
static void se_intra_mode(Enc *e, int pred, int mode) {
    if (!e->cabac) { if (mode == pred) bw_put(&e->bw, 1, 1); else { bw_put(&e->bw, 1, 0); bw_put(&e->bw, 3, (uint32_t)(mode < pred ? mode : mode - 1)); }
        return; }
    CabEnc *c = &e->cab;
    cab_enc(c, 68, mode == pred);
    if (mode != pred) { int rem = mode < pred ? mode : mode - 1; cab_enc(c, 69, rem & 1); cab_enc(c, 69, (rem >> 1) & 1); cab_enc(c, 69, rem >> 2); }
}
static void se_chroma_mode(Enc *e, int mx, int my, int cm) {
    if (!e->cabac) { bw_ue(&e->bw, cm); return; }
    MbE *a = mb_avail(e, mx - 1, my), *b = mb_avail(e, mx, my - 1); CabEnc *c = &e->cab;
    cab_enc(c, 64 + (a && a->cmode != 0) + (b && b->cmode != 0), cm > 0);
    if (cm > 0) { cab_enc(c, 67, cm > 1); if (cm > 1) cab_enc(c, 67, cm > 2); }
}
static void se_cbp(Enc *e, int mx, int my, int cbp, int intra) {
    if (!e->cabac) { const uint8_t *tab = intra ? cbp_intra_tab : cbp_inter_tab; int code = 0; for (int i = 0; i < 48; i++) if (tab[i] == cbp) code = i;
        bw_ue(&e->bw, code); return; }
    MbE *a = mb_avail(e, mx - 1, my), *b = mb_avail(e, mx, my - 1); CabEnc *c = &e->cab;
    for (int b8 = 0; b8 < 4; b8++) {
        int ca = (b8 & 1) ? !((cbp >> (b8 - 1)) & 1) : (a ? !((a->cbp >> (b8 + 1)) & 1) : 0);
        int cb = (b8 & 2) ? !((cbp >> (b8 - 2)) & 1) : (b ? !((b->cbp >> (b8 + 2)) & 1) : 0);
        cab_enc(c, 73 + ca + 2 * cb, (cbp >> b8) & 1);
    }
    int cc = cbp >> 4, ca = a && (a->cbp >> 4) != 0, cb = b && (b->cbp >> 4) != 0;
    cab_enc(c, 77 + ca + 2 * cb, cc != 0);
    if (cc) { ca = a && (a->cbp >> 4) == 2; cb = b && (b->cbp >> 4) == 2; cab_enc(c, 81 + ca + 2 * cb, cc == 2); }
}
static void se_dqp(Enc *e, int dqp) {
    if (!e->cabac) { bw_se(&e->bw, dqp); return; }
    int k = dqp > 0 ? 2 * dqp - 1 : -2 * dqp, ctx = 60 + (e->last_dqp ? 1 : 0);
    for (int i = 0; i < k; i++) { cab_enc(&e->cab, ctx, 1); ctx = 60 + (i == 0 ? 2 : 3); }
    cab_enc(&e->cab, ctx, 0);
    e->last_dqp = dqp != 0;
}
/* residual_block_cabac: coef in scan order (maxnum entries); bit = coded_block_flag bit (-1 for 8x8 blocks); fa/fb neighbours' flags (-1 n/a) */
static int cab_block(Enc *e, MbE *m, int cat, int bit, int fa, int fb, const int *coef, int maxnum) {
    static const int cbf_off[5] = {0, 4, 8, 12, 16}, sig_off[5] = {0, 15, 29, 44, 47}, abs_off[5] = {0, 10, 20, 30, 39};
    CabEnc *c = &e->cab; int n = 0, last = -1;
    for (int i = 0; i < maxnum; i++) if (coef[i]) { n++; last = i; }
    if (cat != 5) {
        if (fa < 0) fa = m->intra; if (fb < 0) fb = m->intra;
        cab_enc(c, 85 + cbf_off[cat] + fa + 2 * fb, n != 0);
        if (!n) return 0;
        m->cbf |= 1u << bit;
    }
    /* significant_coeff_flag / last_significant_coeff_flag have a second set of contexts for field-coded blocks (Table 9-34: 277.. / 338..) */
    int sb = cat == 5 ? 402 : (e->field ? 277 : 105) + sig_off[cat], lb = cat == 5 ? 417 : (e->field ? 338 : 166) + sig_off[cat];
    int ab = cat == 5 ? 426 : 227 + abs_off[cat];
    if (cat == 5 && e->field) { fprintf(stderr, "h264gen: field-coded 8x8 block with CABAC\n"); abort(); }
    for (int i = 0; i < maxnum - 1; i++) {
        int si = cat == 5 ? orc_cabac_sig8_inc[i] : cat == 3 ? MIN(i, 2) : i, li = cat == 5 ? orc_cabac_last8_inc[i] : cat == 3 ? MIN(i, 2) : i;
        cab_enc(c, sb + si, coef[i] != 0);
        if (coef[i]) { cab_enc(c, lb + li, i == last); if (i == last) break; }
    }
    int gt1 = 0, eq1 = 0;
    for (int i = last; i >= 0; i--) {
        if (!coef[i]) continue;
        int v = ABS(coef[i]) - 1;
        cab_enc(c, ab + (gt1 ? 0 : MIN(4, 1 + eq1)), v > 0);
        if (v > 0) {
            int ctx = ab + 5 + MIN(4 - (cat == 3), gt1);
            for (int k = 1; k < MIN(v, 14); k++) cab_enc(c, ctx, 1);
            if (v < 14) cab_enc(c, ctx, 0); else cab_ueg(c, v - 14, 0);
        }
        if (v == 0) eq1++; else gt1++;
        cab_byp(c, coef[i] < 0);
    }
    return n;
}

static void write_mb_residual(Enc *e, int mx, int my, MbE *m, const MbCode *mc) {
    BitW *w = &e->bw; int sc[64];
    MbE *mA = mb_avail(e, mx - 1, my), *mB = mb_avail(e, mx, my - 1);
    if (mc->type == 6) {
        for (int i = 0; i < 16; i++) sc[i] = mc->dc16[e->scan4[i]];
        if (e->cabac) cab_block(e, m, 0, 16, mA ? (int)((mA->cbf >> 16) & 1) : -1, mB ? (int)((mB->cbf >> 16) & 1) : -1, sc, 16);
        else write_block(w, sc, 16, nC_luma(e, mx, my, m, 0, 0));
    }
    for (int b8 = 0; b8 < 4; b8++) {
        if (mc->t8) {
            int ox = (b8 & 1) * 2, oy = (b8 >> 1) * 2, total = 0;
            if (!(mc->cbp & (1 << b8))) { for (int k = 0; k < 4; k++) m->tc[(oy + (k >> 1)) * 4 + ox + (k & 1)] = 0; continue; }
            if (e->cabac) {
                for (int i = 0; i < 64; i++) sc[i] = mc->luma8[b8][e->scan8[i]];
                total = cab_block(e, m, 5, -1, 0, 0, sc, 64);
                for (int k = 0; k < 4; k++) { int r = (oy + (k >> 1)) * 4 + ox + (k & 1); m->tc[r] = (uint8_t)MIN(total, 16); m->cbf |= 1u << r; }
            } else for (int k = 0; k < 4; k++) {
                int bx = ox + (k & 1), by = oy + (k >> 1);
                for (int i = 0; i < 16; i++) sc[i] = mc->luma8[b8][e->scan8[4 * i + k]];
                int n = write_block(w, sc, 16, nC_luma(e, mx, my, m, bx, by));
                m->tc[by * 4 + bx] = (uint8_t)n; total += n;
            }
            if (total) for (int k = 0; k < 4; k++) m->nzmask |= (uint16_t)(1u << ((oy + (k >> 1)) * 4 + ox + (k & 1)));
            continue;
        }
        for (int k = 0; k < 4; k++) {
            int blk = b8 * 4 + k, bx = bX(blk), by = bY(blk), r = by * 4 + bx, n, fa = -1, fb = -1;
            if (!(mc->cbp & (1 << b8))) { m->tc[r] = 0; continue; }
            if (e->cabac) { int q; MbE *nn = nb4(e, mx, my, m, bx, by, 1, &q); if (nn) fa = (int)((nn->cbf >> q) & 1); nn = nb4(e, mx, my, m, bx, by, 0, &q);
                if (nn) fb = (int)((nn->cbf >> q) & 1); }
            if (mc->type == 6) { for (int i = 0; i < 15; i++) sc[i] = mc->luma[r][e->scan4[i + 1]];
                n = e->cabac ? cab_block(e, m, 1, r, fa, fb, sc, 15) : write_block(w, sc, 15, nC_luma(e, mx, my, m, bx, by)); }
            else { for (int i = 0; i < 16; i++) sc[i] = mc->luma[r][e->scan4[i]]; n = e->cabac ? cab_block(e, m, 2, r, fa, fb, sc, 16) : write_block(w, sc, 16,
                nC_luma(e, mx, my, m, bx, by)); }
            m->tc[r] = (uint8_t)n;
            if (n) m->nzmask |= (uint16_t)(1u << r);
        }
    }
    if (mc->cbp & 0x30) for (int pl = 0; pl < 2; pl++) {
        if (e->cabac) cab_block(e, m, 3, 17 + pl, mA ? (int)((mA->cbf >> (17 + pl)) & 1) : -1, mB ? (int)((mB->cbf >> (17 + pl)) & 1) : -1, mc->cdc[pl], 4);
        else write_block(w, mc->cdc[pl], 4, -1);
    }
    for (int pl = 0; pl < 2; pl++) for (int k = 0; k < 4; k++) {
        if (!(mc->cbp & 0x20)) { m->tc[16 + 4 * pl + k] = 0; continue; }
        for (int i = 0; i < 15; i++) sc[i] = mc->cac[pl][k][e->scan4[i + 1]];
        if (e->cabac) {
            int bx = k & 1, by = k >> 1, b0 = 19 + pl * 4, fa, fb;
            if (bx) fa = (int)((m->cbf >> (b0 + by * 2)) & 1); else fa = mA ? (int)((mA->cbf >> (b0 + by * 2 + 1)) & 1) : -1;
            if (by) fb = (int)((m->cbf >> (b0 + bx)) & 1); else fb = mB ? (int)((mB->cbf >> (b0 + 2 + bx)) & 1) : -1;
            m->tc[16 + 4 * pl + k] = (uint8_t)cab_block(e, m, 4, b0 + k, fa, fb, sc, 15);
        } else m->tc[16 + 4 * pl + k] = (uint8_t)write_block(w, sc, 15, nC_chroma(e, mx, my, m, pl, k & 1, k >> 1));
    }
}

/* Intra4x4 predicted mode (8.3.1.1) */
static int i4_pred_mode(Enc *e, int mx, int my, MbE *m, int bx, int by) {
    MbE *mA = bx > 0 ? m : mb_avail(e, mx - 1, my), *mB = by > 0 ? m : mb_avail(e, mx, my - 1);
    if (!mA || !mB) return 2;
    if (e->p.cip && (!mA->intra || !mB->intra)) return 2;
    int a = (mA->intra && !mA->i16 && !mA->pcm) ? (bx > 0 ? m->i4[by * 4 + bx - 1] : mA->i4[by * 4 + 3]) : 2;
    int b = (mB->intra && !mB->i16 && !mB->pcm) ? (by > 0 ? m->i4[(by - 1) * 4 + bx] : mB->i4[12 + bx]) : 2;
    return MIN(a, b);
}

static void mb_init(Enc *e, MbE *m) {
    memset(m, 0, sizeof *m);
    m->slice = e->slice_id; for (int i = 0; i < 4; i++) { m->ref[i] = -1; m->refid[i] = -1; m->ref1[i] = -1; m->refid1[i] = -1; }
    memset(m->i4, 2, 16);
    m->dis_db = (int8_t)(e->p.deblock == 1 ? 0 : (e->p.deblock == 0 ? 1 : 2)); m->a_off = (int8_t)(2 * e->p.alpha_off); m->b_off = (int8_t)(2 * e->p.beta_off);
    e->decoded_mask = 0;
}

/* encode one intra MB (decision + recon + syntax). in P slices mb_type is offset by 5 */
static void encode_intra_mb(Enc *e, int mx, int my, MbE *m, int force /*-1 auto,5 I4,6 I16,7 PCM*/) {
    Frame *c = &e->cur, *s = &e->src; BitW *w = &e->bw; int fuzz = e->p.mode == 1;
    uint8_t *dy = c->y + my * 16 * c->sy + mx * 16;
    m->intra = 1;
    if (force == 7) {
        m->pcm = 1; m->qp = 0; m->qpc[0] = m->qpc[1] = (uint8_t)chroma_qp_of(e, 0);
        se_mb_type_intra(e, mx, my, 25);                      /* CABAC: the terminate bin flushes the arithmetic code (9.3.4.5) */
        while (w->nbits) bw_put(w, 1, 0);
        for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) { int v = s->y[(my * 16 + y) * s->sy + mx * 16 + x]; dy[y * c->sy + x] = (uint8_t)v;
            bw_put(w, 8, v); }
        for (int pl = 0; pl < 2; pl++) for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
            int v = (pl ? s->v : s->u)[(my * 8 + y) * s->sc + mx * 8 + x]; (pl ? c->v : c->u)[(my * 8 + y) * c->sc + mx * 8 + x] = (uint8_t)v; bw_put(w, 8, v);
                }
        memset(m->tc, 16, 24);
        m->cbp = 0x2f; m->cbf = 0x7FFFFFF; e->last_dqp = 0;
        if (e->cabac) cab_start(&e->cab, w);
        return;
    }
    int aA = intra_ok(e, mb_avail(e, mx - 1, my)), aB = intra_ok(e, mb_avail(e, mx, my - 1)), aD = intra_ok(e, mb_avail(e, mx - 1, my - 1));
    static __thread MbCode mc; memset(&mc, 0, sizeof mc);
    /* QP for this MB */
    int dqp = 0;
    if (fuzz && rnd_n(&e->rng, 6) == 0) dqp = rnd_n(&e->rng, 9) - 4;
    int qp = CLIP3(10, 48, e->qp_run + dqp); dqp = qp - e->qp_run;
    /* ---- Intra16x16 candidate ---- */
    int best16 = -1, best16sad = 1 << 30, p16[256];
    { int cand[4], nc = 0; cand[nc++] = 2; if (aB) cand[nc++] = 0; if (aA) cand[nc++] = 1; if (aA && aB && aD) cand[nc++] = 3;
      if (fuzz) { best16 = cand[rnd_n(&e->rng, nc)]; }
      else for (int i = 0; i < nc; i++) { big_predict(dy, c->sy, 16, cand[i], aA, aB, p16); int sd = sad16_pred(e, mx, my, p16); if (sd < best16sad) {
          best16sad = sd; best16 = cand[i]; } } }
    /* ---- decide I4x4 vs I16x16 ---- */
    int use_i4;
    if (force == 5) use_i4 = 1; else if (force == 6) use_i4 = 0;
    else if (fuzz) use_i4 = rnd_n(&e->rng, 2);
    else {   /* estimate I4x4 cost with source neighbours replaced by recon progressively: do a trial encode below and compare SAD */
        use_i4 = best16sad > 16 * 16 * 3;
    }
    int cbp_l = 0;
    if (use_i4) {
        mc.type = 5;
        se_mb_type_intra(e, mx, my, 0);
        int use_i8 = e->p.t8x8 && rnd_n(&e->rng, 2);
        if (e->p.t8x8) se_t8_flag(e, mx, my, use_i8);
        mc.t8 = use_i8; m->t8 = (uint8_t)use_i8;
        if (use_i8) {
            for (int b8 = 0; b8 < 4; b8++) {
                int bx = (b8 & 1) * 2, by = (b8 >> 1) * 2, Tb[17], Lb[9], *T = Tb + 1, *L = Lb + 1, a, b, d, p[64];
                i8_edges(e, mx, my, b8, T, L, &a, &b, &d);
                int best = 2, bests = 1 << 30;
                if (fuzz) { do best = rnd_n(&e->rng, 9); while (!i8_mode_ok(best, a, b, d)); }
                else for (int md = 0; md < 9; md++) if (i8_mode_ok(md, a, b, d)) {
                    i8_predict(md, T, L, a, b, p); int sd = 0;
                    for (int k = 0; k < 64; k++) sd += ABS(s->y[(my * 16 + by * 4 + (k >> 3)) * s->sy + mx * 16 + bx * 4 + (k & 7)] - p[k]);
                    if (sd < bests) { bests = sd; best = md; } }
                i8_predict(best, T, L, a, b, p);
                uint8_t *dd = dy + by * 4 * c->sy + bx * 4;
                for (int k = 0; k < 64; k++) dd[(k >> 3) * c->sy + (k & 7)] = (uint8_t)p[k];
                int pm = i4_pred_mode(e, mx, my, m, bx, by);           /* neighbours' modes only: known before this block's own mode is stored */
                mc.i4modes[b8] = best; mc.i4modes[4 + b8] = pm;
                m->i4[by * 4 + bx] = m->i4[by * 4 + bx + 1] = m->i4[by * 4 + bx + 4] = m->i4[by * 4 + bx + 5] = (uint8_t)best;
                if (code_luma8(e, mx * 16 + bx * 4, my * 16 + by * 4, qp, 1, mc.luma8[b8])) cbp_l |= 1 << b8;
            }
            for (int b8 = 0; b8 < 4; b8++) se_intra_mode(e, mc.i4modes[4 + b8], mc.i4modes[b8]);
        } else {
        /* choose, reconstruct and remember modes; syntax needs all modes before residual so buffer decisions */
        for (int blk = 0; blk < 16; blk++) {
            int bx = bX(blk), by = bY(blk), r = by * 4 + bx, Tb[9], Lb[5], *T = Tb + 1, *L = Lb + 1, a, b, p[16];
            i4_edges(e, mx, my, blk, T, L, &a, &b);
            int best = 2, bests = 1 << 30;
            if (fuzz) { do best = rnd_n(&e->rng, 9); while (!i4_mode_ok(best, a, b)); }
            else for (int md = 0; md < 9; md++) if (i4_mode_ok(md, a, b)) {
                i4_predict(md, T, L, a, b, p); int sd = 0;
                for (int k = 0; k < 16; k++) sd += ABS(s->y[(my * 16 + by * 4 + (k >> 2)) * s->sy + mx * 16 + bx * 4 + (k & 3)] - p[k]);
                if (sd < bests) { bests = sd; best = md; } }
            i4_predict(best, T, L, a, b, p);
            uint8_t *d = dy + by * 4 * c->sy + bx * 4;
            for (int k = 0; k < 16; k++) d[(k >> 2) * c->sy + (k & 3)] = (uint8_t)p[k];
            mc.i4modes[r] = best; m->i4[r] = (uint8_t)best;
            if (code_luma4(e, mx * 16 + bx * 4, my * 16 + by * 4, qp, 1, mc.luma[r])) cbp_l |= 1 << (blk >> 2);
        }
        /* blocks in an 8x8 whose cbp bit is clear must have been all-zero: true by construction of cbp_l */
        for (int blk = 0; blk < 16; blk++) {
            int bx = bX(blk), by = bY(blk), r = by * 4 + bx, pm = i4_pred_mode(e, mx, my, m, bx, by), md = mc.i4modes[r];
            /* i4_pred_mode reads m->i4 of already-coded blocks only (left/top), all set above */
            se_intra_mode(e, pm, md);
        }
        }
    } else {
        mc.type = 6; m->i16 = 1; mc.i16mode = best16;
        big_predict(dy, c->sy, 16, best16, aA, aB, p16);
        for (int k = 0; k < 256; k++) dy[(k >> 4) * c->sy + (k & 15)] = (uint8_t)p16[k];
        int dcw[16], any_ac = 0, wblk[16][16];
        int shift = 15 + qp / 6, f = (1 << shift) / 3;
        for (int r = 0; r < 16; r++) {
            int x[16]; int px = mx * 16 + (r & 3) * 4, py = my * 16 + (r >> 2) * 4;
            for (int k = 0; k < 16; k++) x[k] = s->y[(py + (k >> 2)) * s->sy + px + (k & 3)] - c->y[(py + (k >> 2)) * c->sy + px + (k & 3)];
            fdct4(x, wblk[r]); dcw[r] = wblk[r][0]; mc.luma[r][0] = 0;
            for (int k = 1; k < 16; k++) { mc.luma[r][k] = quant1(wblk[r][k], quant_mf[qp % 6][pos_class(k)] * 16 / g_w4[0][k], f, shift);
                any_ac |= mc.luma[r][k] != 0; }
        }
        /* forward 4x4 Hadamard of the DCs, /2, quantise */
        int t[16], h[16];
        for (int i = 0; i < 4; i++) { int *r = dcw + 4 * i; t[4 * i] = r[0] + r[1] + r[2] + r[3]; t[4 * i + 1] = r[0] + r[1] - r[2] - r[3];
            t[4 * i + 2] = r[0] - r[1] - r[2] + r[3]; t[4 * i + 3] = r[0] - r[1] + r[2] - r[3]; }
        for (int j = 0; j < 4; j++) { h[j] = (t[j] + t[4 + j] + t[8 + j] + t[12 + j]) >> 1; h[4 + j] = (t[j] + t[4 + j] - t[8 + j] - t[12 + j]) >> 1;
            h[8 + j] = (t[j] - t[4 + j] - t[8 + j] + t[12 + j]) >> 1; h[12 + j] = (t[j] - t[4 + j] + t[8 + j] - t[12 + j]) >> 1; }
        for (int k = 0; k < 16; k++) mc.dc16[k] = quant1(h[k], quant_mf[qp % 6][0] * 16 / g_w4[0][0], 2 * f, shift + 1);
        if (!any_ac) for (int r = 0; r < 16; r++) for (int k = 1; k < 16; k++) mc.luma[r][k] = 0;
        cbp_l = any_ac ? 15 : 0;
        /* reconstruct: inverse Hadamard + scaling of DCs (8.5.10) */
        int g[16]; const int *cq = mc.dc16;
        for (int i = 0; i < 4; i++) { const int *r = cq + 4 * i; t[4 * i] = r[0] + r[1] + r[2] + r[3]; t[4 * i + 1] = r[0] + r[1] - r[2] - r[3];
            t[4 * i + 2] = r[0] - r[1] - r[2] + r[3]; t[4 * i + 3] = r[0] - r[1] + r[2] - r[3]; }
        for (int j = 0; j < 4; j++) { g[j] = t[j] + t[4 + j] + t[8 + j] + t[12 + j]; g[4 + j] = t[j] + t[4 + j] - t[8 + j] - t[12 + j];
            g[8 + j] = t[j] - t[4 + j] - t[8 + j] + t[12 + j]; g[12 + j] = t[j] - t[4 + j] + t[8 + j] - t[12 + j]; }
        int ls0 = g_w4[0][0] * norm4[qp % 6][0]; g_wlist = 0;
        for (int r = 0; r < 16; r++) {
            int dq[16];
            dq[0] = qp >= 36 ? (g[r] * ls0) << (qp / 6 - 6) : (g[r] * ls0 + (1 << (5 - qp / 6))) >> (6 - qp / 6);
            for (int k = 1; k < 16; k++) dq[k] = dequant_ac(mc.luma[r][k], qp, k);
            idct4_add(dq, dy + (r >> 2) * 4 * c->sy + (r & 3) * 4, c->sy);
        }
    }
    /* ---- chroma ---- */
    int cand[4], nc = 0, cmode = 0; cand[nc++] = 0; if (aA) cand[nc++] = 1; if (aB) cand[nc++] = 2; if (aA && aB && aD) cand[nc++] = 3;
    if (fuzz) cmode = cand[rnd_n(&e->rng, nc)];
    else { int bests = 1 << 30; for (int i = 0; i < nc; i++) { int sd = 0, pu[64], pv[64], kind = cand[i] == 0 ? 2 : cand[i] == 1 ? 1 : cand[i] == 2 ? 0 : 3;
            big_predict(c->u + my * 8 * c->sc + mx * 8, c->sc, 8, kind, aA, aB, pu); big_predict(c->v + my * 8 * c->sc + mx * 8, c->sc, 8, kind, aA, aB, pv);
            for (int k = 0; k < 64; k++) sd +=
                ABS(s->u[(my * 8 + (k >> 3)) * s->sc + mx * 8 + (k & 7)] - pu[k]) + ABS(s->v[(my * 8 + (k >> 3)) * s->sc + mx * 8 + (k & 7)] - pv[k]);
            if (sd < bests) { bests = sd; cmode = cand[i]; } } }
    int qpc = chroma_qp_of(e, qp), cflags = 0;
    for (int pl = 0; pl < 2; pl++) { int pc[64], kind = cmode == 0 ? 2 : cmode == 1 ? 1 : cmode == 2 ? 0 : 3;
        uint8_t *cp = (pl ? c->v : c->u) + my * 8 * c->sc + mx * 8;
        big_predict(cp, c->sc, 8, kind, aA, aB, pc); for (int k = 0; k < 64; k++) cp[(k >> 3) * c->sc + (k & 7)] = (uint8_t)pc[k]; }
    for (int pl = 0; pl < 2; pl++) cflags |= code_chroma(e, mx, my, pl, qpc, 1, &mc);
    int cbp_c = (cflags & 2) ? 2 : (cflags & 1) ? 1 : 0;
    for (int pl = 0; pl < 2; pl++) recon_chroma(e, mx, my, pl, qpc, &mc, cbp_c >= 1, cbp_c == 2);
    mc.cbp = cbp_l | (cbp_c << 4); mc.cmode = cmode;
    /* ---- syntax ---- */
    if (mc.type == 6) se_mb_type_intra(e, mx, my, 1 + mc.i16mode + 4 * cbp_c + (cbp_l ? 12 : 0));
    se_chroma_mode(e, mx, my, cmode); m->cmode = (uint8_t)cmode;
    if (mc.type == 5) se_cbp(e, mx, my, mc.cbp, 1);
    m->cbp = (uint8_t)mc.cbp;
    if (mc.cbp > 0 || mc.type == 6) { se_dqp(e, dqp); e->qp_run = qp; } else { qp = e->qp_run; e->last_dqp = 0; }
    m->qp = (uint8_t)qp; m->qpc[0] = m->qpc[1] = (uint8_t)chroma_qp_of(e, qp);
    if (mc.cbp > 0 || mc.type == 6) write_mb_residual(e, mx, my, m, &mc);
    /* note: when cbp==0 for I4x4 we quantised with a QP that is never signalled, but all levels are zero so recon is unaffected */
}

typedef struct { int mvx, mvy, cost; } MvRes;
static int mv_bits(int d) { int a = ABS(d) * 2 + 1, n = 0; while (a >> n) n++; return 2 * n - 1; }
static MvRes search_block(Enc *e, const Frame *r, int px, int py, int w, int h, const int mvp[2], int cx, int cy, int range) {
    MvRes qbest = { 0, 0, 1 << 30 }; int have_q = 0;
    MvRes best = { 0, 0, 1 << 30 }; int lambda = 4;
    if (mv_legal(e, px, py, w, h, mvp[0], mvp[1])) {          /* constant-velocity content: the predictor is usually exact */
        int c = sad_inter(e, r, px, py, w, h, mvp[0], mvp[1]);
        if (c < w * h * 5 / 2) { best.mvx = mvp[0]; best.mvy = mvp[1]; best.cost = c; return best; }
        best.mvx = mvp[0]; best.mvy = mvp[1]; best.cost = c;
    }
    int cands[3][2] = { { mvp[0] & ~3, mvp[1] & ~3 }, { 0, 0 }, { cx & ~3, cy & ~3 } };
    for (int i = 0; i < 3; i++) { int mx = cands[i][0], my = cands[i][1]; if (!mv_legal(e, px, py, w, h, mx, my)) continue;
        int c = sad_inter(e, r, px, py, w, h, mx, my) + lambda * (mv_bits(mx - mvp[0]) + mv_bits(my - mvp[1])); if (c < best.cost) { best.mvx = mx;
            best.mvy = my; best.cost = c; } }
    if (best.cost == 1 << 30) { best.mvx = best.mvy = 0; best.cost = sad_inter(e, r, px, py, w, h, 0, 0); }
    best.mvx &= ~3; best.mvy &= ~3;
    best.cost = sad_inter(e, r, px, py, w, h, best.mvx, best.mvy) + lambda * (mv_bits(best.mvx - mvp[0]) + mv_bits(best.mvy - mvp[1]));
    { int c = sad_inter(e, r, px, py, w, h, mvp[0], mvp[1]); if (mv_legal(e, px, py, w, h, mvp[0], mvp[1]) && c <= best.cost) { MvRes q = {
        mvp[0], mvp[1], c }; qbest = q; have_q = 1; } }
    for (int step = range; step >= 1; step >>= 1) {          /* integer: shrinking-step pattern search */
        int improved = 1;
        while (improved) { improved = 0; int bx = best.mvx, by = best.mvy;
            static const int dx[8] = { -1, 1, 0, 0, -1, 1, -1, 1 }, dy[8] = { 0, 0, -1, 1, -1, -1, 1, 1 };
            for (int k = 0; k < 8; k++) { int mx = bx + dx[k] * step * 4, my = by + dy[k] * step * 4; if (!mv_legal(e, px, py, w, h, mx, my)) continue;
                int c = sad_inter(e, r, px, py, w, h, mx, my) + lambda * (mv_bits(mx - mvp[0]) + mv_bits(my - mvp[1])); if (c < best.cost) { best.mvx = mx;
                    best.mvy = my; best.cost = c; improved = 1; } } }
    }
    for (int step = 2; step >= 1; step--) {                  /* half then quarter */
        int bx = best.mvx, by = best.mvy;
        for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) { if (!dx && !dy) continue; int mx = bx + dx * step, my = by + dy * step;
            if (!mv_legal(e, px, py, w, h, mx, my)) continue;
            int c = sad_inter(e, r, px, py, w, h, mx, my) + lambda * (mv_bits(mx - mvp[0]) + mv_bits(my - mvp[1])); if (c < best.cost) { best.mvx = mx;
                best.mvy = my; best.cost = c; } }
    }
    if (have_q && qbest.cost <= best.cost) return qbest;
    return best;
}
static void random_mv(Enc *e, int px, int py, int w, int h, const int mvp[2], int out[2]) {
    for (int tries = 0; tries < 50; tries++) {
        int k = rnd_n(&e->rng, 10), mx, my;
        if (k < 4) { mx = mvp[0] + rnd_n(&e->rng, 9) - 4; my = mvp[1] + rnd_n(&e->rng, 9) - 4; }
        else if (k < 8) { mx = rnd_n(&e->rng, 65) - 32; my = rnd_n(&e->rng, 65) - 32; }
        else { mx = rnd_n(&e->rng, 4 * 161) - 4 * 80; my = rnd_n(&e->rng, 4 * 161) - 4 * 80; }
        if (mv_legal(e, px, py, w, h, mx, my)) { out[0] = mx; out[1] = my; return; }
    }
    out[0] = out[1] = 0;
}

static void mc_mb(Enc *e, int mx, int my, MbE *m);
static void encode_p_mb(Enc *e, int mx, int my, MbE *m, int *skip_run) {
    Frame *c = &e->cur; BitW *w = &e->bw; int fuzz = e->p.mode == 1;
    int px = mx * 16, py = my * 16, nref = e->nlist0;
    const int r0 = e->n_usable ? e->usable[0] : 0;                      /* the first entry of list 0 that exists (gaps in frame_num: GenParams.gaps) */
    /* ---- decide intra vs inter ---- */
    int want_intra = 0, force_intra = -1;
    if (fuzz) { int k = rnd_n(&e->rng, 16); if (k == 0) { want_intra = 1; force_intra = -1; } else if (k == 1 && rnd_n(&e->rng, 4) == 0) { want_intra = 1;
        force_intra = 7; } if (e->p.no_intra) want_intra = 0; }
    static __thread MbCode mc; memset(&mc, 0, sizeof mc);
    int try_skip = fuzz && !want_intra && rnd_n(&e->rng, 8) < 2;
    if (r0 != 0) try_skip = 0;                                           /* P_Skip predicts from entry 0 */
    int type = 0, sub[4] = {0, 0, 0, 0}, refs[4] = {r0, r0, r0, r0};
    int mvs[16][2]; memset(mvs, 0, sizeof mvs);
    int skipmv[2]; skip_mv(e, mx, my, m, skipmv);
    if (!want_intra) {
        if (fuzz) {
            type = rnd_n(&e->rng, 4);
            for (int i = 0; i < 4; i++) { sub[i] = rnd_n(&e->rng, 4); refs[i] = nref > 1 ? rnd_n(&e->rng, nref) : 0;
                if (e->n_usable < e->nlist0) refs[i] = e->usable[refs[i] % e->n_usable]; }
            if (type == 0) refs[1] = refs[2] = refs[3] = refs[0];
            else if (type == 1) { refs[1] = refs[0]; refs[3] = refs[2]; }
            else if (type == 2) { refs[2] = refs[0]; refs[3] = refs[1]; }
            if (try_skip) { type = 0; refs[0] = refs[1] = refs[2] = refs[3] = 0; }
        } else {
            int mvp[2]; pred_mv(e, mx, my, m, 0, 0, 4, r0, 0, 0, mvp);
            MvRes r16 = search_block(e, e->list0[r0], px, py, 16, 16, mvp, skipmv[0], skipmv[1], e->p.search);
            for (int k = 0; k < 16; k++) { mvs[k][0] = r16.mvx; mvs[k][1] = r16.mvy; }
            type = 0;
            if (r16.cost > 16 * 16 * 4) {
                /* try an 8x8 split around the 16x16 vector */
                MvRes r8[4]; int tot = 0, c8[2] = { r16.mvx, r16.mvy };
                for (int i = 0; i < 4; i++) { r8[i] = search_block(e, e->list0[r0], px + (i & 1) * 8, py + (i >> 1) * 8, 8, 8, c8, r16.mvx, r16.mvy, 2);
                    tot += r8[i].cost; }
                if (tot + 16 * 12 < r16.cost) {
                    int same_h = r8[0].mvx == r8[1].mvx && r8[0].mvy == r8[1].mvy && r8[2].mvx == r8[3].mvx && r8[2].mvy == r8[3].mvy;
                    int same_v = r8[0].mvx == r8[2].mvx && r8[0].mvy == r8[2].mvy && r8[1].mvx == r8[3].mvx && r8[1].mvy == r8[3].mvy;
                    type = same_h ? 1 : same_v ? 2 : 3;
                    for (int i = 0; i < 4; i++) {
                        int bx = (i & 1) * 2, by = (i >> 1) * 2;
                        for (int k = 0; k < 4; k++) { mvs[(by + (k >> 1)) * 4 + bx + (k & 1)][0] = r8[i].mvx;
                            mvs[(by + (k >> 1)) * 4 + bx + (k & 1)][1] = r8[i].mvy; }
                        if (type == 3 && r8[i].cost > 8 * 8 * 6) {     /* 4x4 split of a still-poor 8x8 */
                            int c4[2] = { r8[i].mvx, r8[i].mvy }, t4 = 0; MvRes r4[4];
                            for (int k = 0; k < 4; k++) { r4[k] = search_block(e, e->list0[r0], px + bx * 4 + (k & 1) * 4, py + by * 4 + (k >> 1) * 4, 4, 4, c4,
                                c4[0], c4[1], 1); t4 += r4[k].cost; }
                            if (t4 + 40 < r8[i].cost) {
                                int sh = r4[0].mvx == r4[1].mvx && r4[0].mvy == r4[1].mvy && r4[2].mvx == r4[3].mvx && r4[2].mvy == r4[3].mvy;
                                int sv = r4[0].mvx == r4[2].mvx && r4[0].mvy == r4[2].mvy && r4[1].mvx == r4[3].mvx && r4[1].mvy == r4[3].mvy;
                                sub[i] = sh ? 1 : sv ? 2 : 3;
                                for (int k = 0; k < 4; k++) { mvs[(by + (k >> 1)) * 4 + bx + (k & 1)][0] = r4[k].mvx;
                                    mvs[(by + (k >> 1)) * 4 + bx + (k & 1)][1] = r4[k].mvy; }
                            }
                        }
                    }
                }
            }
            /* intra fallback when inter prediction is poor */
            int best_cost = 0;
            for (int k = 0; k < 16; k++) best_cost += sad_inter(e, e->list0[r0], px + (k & 3) * 4, py + (k >> 2) * 4, 4, 4, mvs[k][0], mvs[k][1]);
            if (best_cost > 16 * 16 * 10) {
                int aA = intra_ok(e, mb_avail(e, mx - 1, my)), aB = intra_ok(e, mb_avail(e, mx, my - 1)), p16[256];
                big_predict(c->y + py * c->sy + px, c->sy, 16, 2, aA, aB, p16);
                if (sad16_pred(e, mx, my, p16) < best_cost) want_intra = 1;
            }
        }
    }
    if (want_intra) {
        se_begin_mb(e, mx, my, skip_run);
        encode_intra_mb(e, mx, my, m, force_intra);
        return;
    }
    /* ---- finalise motion: walk partitions in syntax order, computing mvd against the running prediction ---- */
    int mvd[16][2], mvdpos[16][4], nmvd = 0;
    e->decoded_mask = 0;
    for (int i = 0; i < 4; i++) m->ref[i] = (int8_t)refs[i];
    if (type <= 2) {
        int np = type == 0 ? 1 : 2;
        for (int p = 0; p < np; p++) {
            int bx = type == 2 ? p * 2 : 0, by = type == 1 ? p * 2 : 0, bw = type == 2 ? 2 : 4, bh = type == 1 ? 2 : 4;
            int ref = refs[(by >> 1) * 2 + (bx >> 1)], mvp[2], mv[2];
            pred_mv(e, mx, my, m, bx, by, bw, ref, type, p, mvp);
            if (try_skip && mv_legal(e, px, py, 16, 16, skipmv[0], skipmv[1])) { mv[0] = skipmv[0]; mv[1] = skipmv[1]; }
            else if (fuzz) random_mv(e, px + bx * 4, py + by * 4, bw * 4, bh * 4, mvp, mv); else { mv[0] = mvs[by * 4 + bx][0]; mv[1] = mvs[by * 4 + bx][1]; }
            mvdpos[nmvd][0] = bx; mvdpos[nmvd][1] = by; mvdpos[nmvd][2] = bw; mvdpos[nmvd][3] = bh;
            mvd[nmvd][0] = mv[0] - mvp[0]; mvd[nmvd][1] = mv[1] - mvp[1]; nmvd++;
            store_mv(e, m, bx, by, bw, bh, mv[0], mv[1]);
        }
    } else {
        for (int i = 0; i < 4; i++) {
            int ox = (i & 1) * 2, oy = (i >> 1) * 2, st = sub[i], nsp = st == 0 ? 1 : st == 3 ? 4 : 2, bw = (st == 0 || st == 1) ? 2 : 1,
                bh = (st == 0 || st == 2) ? 2 : 1;
            for (int p = 0; p < nsp; p++) {
                int bx = ox + (st == 1 ? 0 : st == 2 ? p : (p & 1)), by = oy + (st == 1 ? p : st == 2 ? 0 : (p >> 1)), mvp[2], mv[2];
                pred_mv(e, mx, my, m, bx, by, bw, refs[i], 0, 0, mvp);
                if (fuzz) random_mv(e, px + bx * 4, py + by * 4, bw * 4, bh * 4, mvp, mv); else { mv[0] = mvs[by * 4 + bx][0]; mv[1] = mvs[by * 4 + bx][1]; }
                mvdpos[nmvd][0] = bx; mvdpos[nmvd][1] = by; mvdpos[nmvd][2] = bw; mvdpos[nmvd][3] = bh;
                mvd[nmvd][0] = mv[0] - mvp[0]; mvd[nmvd][1] = mv[1] - mvp[1]; nmvd++;
                store_mv(e, m, bx, by, bw, bh, mv[0], mv[1]);
            }
        }
    }
    for (int i = 0; i < 4; i++) m->refid[i] = e->list0[refs[i]]->id;
    /* ---- prediction + residual ---- */
    for (int k = 0; k < 16; k++) mc_block(e, e->list0[refs[(k >> 3) * 2 + ((k & 3) >> 1)]], px + (k & 3) * 4, py + (k >> 2) * 4, 4, 4, m->mv[k][0],
        m->mv[k][1]);
    if (e->p.wp == 1) mc_mb(e, mx, my, m);                               /* explicit weighted prediction (8.4.2.3) */
    int dqp = 0;
    if (fuzz && rnd_n(&e->rng, 6) == 0) dqp = rnd_n(&e->rng, 9) - 4;
    int qp = CLIP3(10, 48, e->qp_run + dqp); dqp = qp - e->qp_run;
    int cbp_l = 0;
    int t8_ok = e->p.t8x8 && (type != 3 || (sub[0] | sub[1] | sub[2] | sub[3]) == 0);     /* noSubMbPartSizeLessThan8x8Flag */
    int use_t8 = t8_ok && (fuzz ? rnd_n(&e->rng, 2) : 1);
    if (!try_skip && use_t8) { for (int b8 = 0; b8 < 4; b8++) if (code_luma8(e, px + (b8 & 1) * 8, py + (b8 >> 1) * 8, qp, 0, mc.luma8[b8])) cbp_l |= 1 << b8; }
    else if (!try_skip) for (int blk = 0; blk < 16; blk++) { int bx = bX(blk), by = bY(blk);
        if (code_luma4(e, px + bx * 4, py + by * 4, qp, 0, mc.luma[by * 4 + bx])) cbp_l |= 1 << (blk >> 2); }
    if (!cbp_l) use_t8 = 0;
    mc.t8 = use_t8;
    int qpc = chroma_qp_of(e, qp), cflags = 0;
    if (!try_skip) for (int pl = 0; pl < 2; pl++) cflags |= code_chroma(e, mx, my, pl, qpc, 0, &mc);
    int cbp_c = (cflags & 2) ? 2 : (cflags & 1) ? 1 : 0;
    for (int pl = 0; pl < 2; pl++) recon_chroma(e, mx, my, pl, qpc, &mc, cbp_c >= 1, cbp_c == 2);
    mc.cbp = cbp_l | (cbp_c << 4); mc.type = type;
    /* ---- P_Skip ---- */
    if (type == 0 && refs[0] == 0 && mc.cbp == 0 && m->mv[0][0] == skipmv[0] && m->mv[0][1] == skipmv[1] && !(fuzz && rnd_n(&e->rng, 4) == 0)) {
        m->skip = 1; m->qp = (uint8_t)e->qp_run; m->qpc[0] = m->qpc[1] = (uint8_t)chroma_qp_of(e, e->qp_run);
        if (e->cabac) { se_skip_flag(e, mx, my, 1); e->last_dqp = 0; } else (*skip_run)++;
        return;
    }
    se_begin_mb(e, mx, my, skip_run);
    int p8ref0 = !e->cabac && type == 3 && nref > 1 && refs[0] == 0 && refs[1] == 0 && refs[2] == 0 && refs[3] == 0 && (fuzz ? rnd_n(&e->rng, 2) : 1);
    se_mb_type_p(e, type, p8ref0);
    /* the CABAC contexts of ref_idx look at m->ref of earlier partitions only; the final values are already stored, which is
       equivalent because partitions later in syntax order are never the left / upper neighbour of an earlier one */
    if (type <= 2) {
        int np = type == 0 ? 1 : 2;
        if (nref > 1) for (int p = 0; p < np; p++) se_ref_idx(e, mx, my, m, type == 2 ? p * 2 : 0, type == 1 ? p * 2 : 0, nref, refs[type == 1 ? p * 2 : p]);
    } else {
        for (int i = 0; i < 4; i++) se_sub_mb_type(e, sub[i]);
        if (nref > 1 && !p8ref0) for (int i = 0; i < 4; i++) se_ref_idx(e, mx, my, m, (i & 1) * 2, (i >> 1) * 2, nref, refs[i]);
    }
    for (int i = 0; i < nmvd; i++) se_mvd(e, mx, my, m, mvdpos[i][0], mvdpos[i][1], mvdpos[i][2], mvdpos[i][3], mvd[i][0], mvd[i][1]);
    se_cbp(e, mx, my, mc.cbp, 0); m->cbp = (uint8_t)mc.cbp;
    if (cbp_l && t8_ok) se_t8_flag(e, mx, my, use_t8);
    m->t8 = (uint8_t)use_t8;
    if (mc.cbp > 0) { se_dqp(e, dqp); e->qp_run = qp; } else { qp = e->qp_run; e->last_dqp = 0; }
    m->qp = (uint8_t)qp; m->qpc[0] = m->qpc[1] = (uint8_t)chroma_qp_of(e, qp);
    if (mc.cbp > 0) write_mb_residual(e, mx, my, m, &mc);
}

/* ------------------------------ B pictures ----------------------------------- */
static int8_t *mb_ref(MbE *m, int l) { return l ? m->ref1 : m->ref; }
static int16_t (*mb_mv(MbE *m, int l))[2] { return l ? m->mv1 : m->mv; }
static Nbr nbr_get_l(Enc *e, int mx, int my, MbE *cur, int l, int bx, int by) {
    Nbr n = {0, -1, {0, 0}}; MbE *m; int rx, ry;
    if (by < 0) { ry = 3;
        if (bx < 0) { m = mb_avail(e, mx - 1, my - 1); rx = 3; }
        else if (bx > 3) { m = mb_avail(e, mx + 1, my - 1); rx = bx - 4; }
        else { m = mb_avail(e, mx, my - 1); rx = bx; } }
    else if (bx < 0) { m = mb_avail(e, mx - 1, my); rx = 3; ry = by; }
    else if (bx > 3) return n;
    else { if (!(e->decoded_mask >> (by * 4 + bx) & 1)) return n; m = cur; rx = bx; ry = by; }
    if (!m) return n;
    n.avail = 1;
    if (m->intra) return n;
    n.ref = mb_ref(m, l)[(ry >> 1) * 2 + (rx >> 1)];
    if (n.ref >= 0) { n.mv[0] = mb_mv(m, l)[ry * 4 + rx][0]; n.mv[1] = mb_mv(m, l)[ry * 4 + rx][1]; }
    return n;
}
static void pred_mv_l(Enc *e, int mx, int my, MbE *cur, int l, int bx, int by, int bw, int ref, int shape, int part, int out[2]) {
    Nbr A = nbr_get_l(e, mx, my, cur, l, bx - 1, by), B = nbr_get_l(e, mx, my, cur, l, bx, by - 1), C = nbr_get_l(e, mx, my, cur, l, bx + bw, by - 1);
    if (!C.avail) C = nbr_get_l(e, mx, my, cur, l, bx - 1, by - 1);
    if (shape == 1) { if (part == 0 && B.ref == ref) { out[0] = B.mv[0]; out[1] = B.mv[1]; return; } if (part == 1 && A.ref == ref) { out[0] = A.mv[0];
        out[1] = A.mv[1]; return; } }
    if (shape == 2) { if (part == 0 && A.ref == ref) { out[0] = A.mv[0]; out[1] = A.mv[1]; return; } if (part == 1 && C.ref == ref) { out[0] = C.mv[0];
        out[1] = C.mv[1]; return; } }
    if (!B.avail && !C.avail && A.avail) { B = A; C = A; }
    int ma = A.ref == ref, mb = B.ref == ref, mc = C.ref == ref;
    if (ma + mb + mc == 1) { Nbr *n = ma ? &A : (mb ? &B : &C); out[0] = n->mv[0]; out[1] = n->mv[1]; }
    else { out[0] = med3(A.mv[0], B.mv[0], C.mv[0]); out[1] = med3(A.mv[1], B.mv[1], C.mv[1]); }
}
static void store_mv_l(Enc *e, MbE *m, int l, int bx, int by, int bw, int bh, int mvx, int mvy) {
    for (int y = by; y < by + bh; y++) for (int x = bx; x < bx + bw; x++) { mb_mv(m, l)[y * 4 + x][0] = (int16_t)mvx; mb_mv(m, l)[y * 4 + x][1] = (int16_t)mvy;
        e->decoded_mask |= 1 << (y * 4 + x); }
}
/* colocated motion (8.4.1.2.1, Tables 8-6 / 8-8) of 4x4 block r of the current macroblock in the first picture of list 1.
   *vscale: 0 = the colocated picture is coded like the current one (One_To_One); 1 = a field picture takes the motion of a FRAME picture (Frm_To_Fld:
   the field macroblock covers two frame macroblocks, vertical vectors are halved for temporal prediction); 2 = a frame picture takes the motion of one
   FIELD of a field pair -- the one nearer in order count -- (Fld_To_Frm: vertical vectors doubled).  Only with direct_8x8_inference (r a corner block).
   *refid names the referenced picture in the current picture's terms: the field of the current parity of that frame (1), the frame of that field (2). */
static void col_motion(Enc *e, int mx, int my, int r, int *refidx, int mv[2], int *refid, int *vscale) {
    const Frame *l1 = e->list1[0];
    const MbE *cm; int rb = r;
    *vscale = 0;
    if (e->field && !l1->store_fields) {                      /* FLD x FRM */
        const int yCol = (r >> 2) * 4, xCol = (r & 3) * 4, yM = (2 * yCol) % 16;
        cm = &((const MbE *)l1->store_mf)[(2 * my + yCol / 8) * e->mbw + mx];
        rb = (yM >> 2) * 4 + (xCol >> 2); *vscale = 1;
    } else if (!e->field && l1->coded_fields) {               /* FRM x FLD */
        const int yCol = (r >> 2) * 4, xCol = (r & 3) * 4, yM = 8 * (my % 2) + 4 * (yCol / 8);
        const int q = ABS(l1->fpoc[0] - e->cur_poc) < ABS(l1->fpoc[1] - e->cur_poc) ? 0 : 1;
        cm = &((const MbE *)l1->fld[q].mf)[(my / 2) * e->mbw + mx];
        rb = (yM >> 2) * 4 + (xCol >> 2); *vscale = 2;
    } else cm = &((const MbE *)l1->mf)[my * e->mbw + mx];
    int b8 = (rb >> 3) * 2 + ((rb & 3) >> 1);
    *refidx = -1; mv[0] = mv[1] = 0; *refid = -1;
    if (cm->intra) return;
    if (cm->ref[b8] >= 0) { *refidx = cm->ref[b8]; mv[0] = cm->mv[rb][0]; mv[1] = cm->mv[rb][1]; *refid = cm->refid[b8]; }
    else if (cm->ref1[b8] >= 0) { *refidx = cm->ref1[b8]; mv[0] = cm->mv1[rb][0]; mv[1] = cm->mv1[rb][1]; *refid = cm->refid1[b8]; }
    if (*refid >= 0 && *vscale == 1) *refid = (1 << 20) + 2 * *refid + (e->field - 1);
    if (*refid >= 0 && *vscale == 2) *refid = (*refid - (1 << 20)) >> 1;
}
/* 8.4.1.2.2 / 8.4.1.2.3: fill refs and motion vectors of the 8x8 quadrants in mask */
static void b_direct(Enc *e, int mx, int my, MbE *m, int mask) {
    int inf8 = e->p.dinf8;
    if (!e->p.direct_temporal) {
        int ref[2], mvp[2][2] = {{0, 0}, {0, 0}}, saved = e->decoded_mask;
        e->decoded_mask = 0;
        for (int l = 0; l < 2; l++) {
            Nbr n[3] = { nbr_get_l(e, mx, my, m, l, -1, 0), nbr_get_l(e, mx, my, m, l, 0, -1), nbr_get_l(e, mx, my, m, l, 4, -1) };
            if (!n[2].avail) n[2] = nbr_get_l(e, mx, my, m, l, -1, -1);
            ref[l] = -1;
            for (int i = 0; i < 3; i++) if (n[i].ref >= 0 && (ref[l] < 0 || n[i].ref < ref[l])) ref[l] = n[i].ref;   /* smallest non-negative reference index */
        }
        int both_missing = ref[0] < 0 && ref[1] < 0;
        if (both_missing) ref[0] = ref[1] = 0;
        else for (int l = 0; l < 2; l++) if (ref[l] >= 0) pred_mv_l(e, mx, my, m, l, 0, 0, 4, ref[l], 0, 0, mvp[l]);
        e->decoded_mask = saved;
        for (int q = 0; q < 4; q++) if (mask & (1 << q)) {
            m->ref[q] = (int8_t)ref[0]; m->ref1[q] = (int8_t)ref[1];
            for (int k = 0; k < 4; k++) {
                int r = ((q >> 1) * 2 + (k >> 1)) * 4 + (q & 1) * 2 + (k & 1);
                int rc = inf8 ? (q >> 1) * 12 + (q & 1) * 3 : r, cref, cmv[2], cid, vs;
                col_motion(e, mx, my, rc, &cref, cmv, &cid, &vs);      /* (colZeroFlag looks at the vectors as they are stored: no vertical scaling) */
                int still = cref == 0 && ABS(cmv[0]) <= 1 && ABS(cmv[1]) <= 1;     /* colZeroFlag (the list-1 picture is never long-term here) */
                for (int l = 0; l < 2; l++) {
                    int zero = both_missing || ref[l] < 0 || (ref[l] == 0 && still);
                    mb_mv(m, l)[r][0] = (int16_t)(zero ? 0 : mvp[l][0]); mb_mv(m, l)[r][1] = (int16_t)(zero ? 0 : mvp[l][1]);
                }
            }
        }
    } else {
        for (int q = 0; q < 4; q++) if (mask & (1 << q)) for (int k = 0; k < 4; k++) {
            int r = ((q >> 1) * 2 + (k >> 1)) * 4 + (q & 1) * 2 + (k & 1);
            int rc = inf8 ? (q >> 1) * 12 + (q & 1) * 3 : r, cref, cmv[2], cid, r0 = 0, vs;
            col_motion(e, mx, my, rc, &cref, cmv, &cid, &vs);
            if (vs == 1) cmv[1] = cmv[1] / 2; else if (vs == 2) cmv[1] *= 2;     /* 8.4.1.2.3: Frm_To_Fld / Fld_To_Frm */
            if (cref >= 0) { r0 = 0; for (int i = e->nlist0 - 1; i >= 0; i--) if (e->list0[i]->id == cid) r0 = i; }
            int pd = e->list1[0]->poc - e->list0[r0]->poc, pb = e->cur_poc - e->list0[r0]->poc;
            int td = CLIP3(-128, 127, pd), tb = CLIP3(-128, 127, pb), v0[2], v1[2];
            if (td == 0) { v0[0] = cmv[0]; v0[1] = cmv[1]; v1[0] = v1[1] = 0; }
            else {
                int tx = (16384 + ABS(td / 2)) / td, scale = CLIP3(-1024, 1023, (tb * tx + 32) >> 6);
                for (int c = 0; c < 2; c++) { v0[c] = (scale * cmv[c] + 128) >> 8; v1[c] = v0[c] - cmv[c]; }
            }
            m->ref[q] = (int8_t)r0; m->ref1[q] = 0;
            m->mv[r][0] = (int16_t)v0[0]; m->mv[r][1] = (int16_t)v0[1]; m->mv1[r][0] = (int16_t)v1[0]; m->mv1[r][1] = (int16_t)v1[1];
        }
    }
    m->direct8 |= (uint8_t)mask;
}
/* temporal direct needs the colocated block's reference in list 0; the generator only emits it when that holds for every block */
static int temporal_direct_ok(Enc *e, int mx, int my) {
    for (int r = 0; r < 16; r++) { int cref, cmv[2], cid, found = 0, vs; col_motion(e, mx, my, r, &cref, cmv, &cid, &vs); if (cref < 0) continue;
        for (int i = 0; i < e->nlist0; i++) if (e->list0[i]->id == cid) found = 1;
        if (!found) return 0; }
    return 1;
}
/* prediction samples of one 4x4 block from one reference (literal, clamped: direct vectors may point anywhere) */
static void sample4(Enc *e, const Frame *r, int px, int py, int mvx, int mvy, int *yl, int *cu, int *cv) {
    {   /* inside the padded half-sample planes the fast fetch gives the same samples (mc_block checks that under H264GEN_CHECK); the literal
         * form -- ~40 clamped reads per sample -- is only needed for vectors that leave them */
        const int lim = PAD - 8, x0 = px + (mvx >> 2), y0 = py + (mvy >> 2);
        if (r->hb && x0 >= -lim && y0 >= -lim && x0 + 4 <= e->W + lim && y0 + 4 <= e->H + lim)
            for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) yl[y * 4 + x] = qpel_fast(r, x0 + x, y0 + y, mvx & 3, mvy & 3);
        else
            for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) yl[y * 4 + x] = luma_sample(r, e->W, e->H, px + x + (mvx >> 2), py + y + (mvy >> 2),
                mvx & 3, mvy & 3);
    }
    if (e->field) mvy += r->parity == (e->field - 1) ? 0 : (r->parity ? -2 : 2);       /* Table 8-9, as in mc_block */
    int cw = e->W / 2, ch = e->H / 2, fx = mvx & 7, fy = mvy & 7;
    for (int pl = 0; pl < 2; pl++) { const uint8_t *rp = pl ? r->v : r->u; int *o = pl ? cv : cu;
        for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) { int xi = px / 2 + x + (mvx >> 3), yi = py / 2 + y + (mvy >> 3);
            int A = refpx(rp, r->sc, cw, ch, xi, yi), B = refpx(rp, r->sc, cw, ch, xi + 1, yi), C = refpx(rp, r->sc, cw, ch, xi, yi + 1),
                D = refpx(rp, r->sc, cw, ch, xi + 1, yi + 1);
            o[y * 2 + x] = ((8 - fx) * (8 - fy) * A + fx * (8 - fy) * B + (8 - fx) * fy * C + fx * fy * D + 32) >> 6; } }
}
/* 8.4.2.3 for one colour component cmp (0 Y, 1 Cb, 2 Cr) of a block predicted from list entries r0 / r1 (-1 = unused) */
static void weigh(Enc *e, int cmp, int r0, int r1, const int *a, const int *b, int n, int *out) {
    int mode = e->slice_type == 0 ? (e->p.wp == 1) : e->p.wp, lg = e->wlog[cmp != 0];
    for (int i = 0; i < n; i++) {
        int v;
        if (r0 >= 0 && r1 >= 0) {
            if (mode == 0) v = (a[i] + b[i] + 1) >> 1;
            else {
                int w0, w1, o = 0, sh = 5;
                if (mode == 1) { w0 = e->ww[0][r0][cmp]; w1 = e->ww[1][r1][cmp]; o = (e->wo[0][r0][cmp] + e->wo[1][r1][cmp] + 1) >> 1; sh = lg; }
                else {
                    int td = CLIP3(-128, 127, e->list1[r1]->poc - e->list0[r0]->poc), tb = CLIP3(-128, 127, e->cur_poc - e->list0[r0]->poc);
                    w0 = w1 = 32;
                    if (td != 0) { int tx = (16384 + ABS(td / 2)) / td, sc = CLIP3(-1024, 1023, (tb * tx + 32) >> 6) >> 2; if (sc >= -64 && sc <= 128) {
                        w1 = sc; w0 = 64 - sc; } }
                }
                v = CLIP1(((a[i] * w0 + b[i] * w1 + (1 << sh)) >> (sh + 1)) + o);
            }
        } else {
            int l = r0 >= 0 ? 0 : 1, r = l ? r1 : r0; const int *p = l ? b : a;
            if (mode == 1) v = CLIP1((lg >= 1 ? ((p[i] * e->ww[l][r][cmp] + (1 << (lg - 1))) >> lg) : p[i] * e->ww[l][r][cmp]) + e->wo[l][r][cmp]);
            else v = p[i];
        }
        out[i] = v;
    }
}
/* motion compensation of the whole macroblock from m's refs / vectors (both lists, weights) into the current frame */
static void mc_mb(Enc *e, int mx, int my, MbE *m) {
    Frame *c = &e->cur;
    for (int r = 0; r < 16; r++) {
        int q = (r >> 3) * 2 + ((r & 3) >> 1), px = mx * 16 + (r & 3) * 4, py = my * 16 + (r >> 2) * 4;
        int r0 = m->ref[q], r1 = m->ref1[q], y0[16], y1[16], u0[4], u1[4], v0[4], v1[4], oy[16], ou[4], ov[4];
        if (r0 >= 0) sample4(e, e->list0[r0], px, py, m->mv[r][0], m->mv[r][1], y0, u0, v0);
        if (r1 >= 0) sample4(e, e->list1[r1], px, py, m->mv1[r][0], m->mv1[r][1], y1, u1, v1);
        weigh(e, 0, r0, r1, y0, y1, 16, oy); weigh(e, 1, r0, r1, u0, u1, 4, ou); weigh(e, 2, r0, r1, v0, v1, 4, ov);
        for (int i = 0; i < 16; i++) c->y[(py + (i >> 2)) * c->sy + px + (i & 3)] = (uint8_t)oy[i];
        for (int i = 0; i < 4; i++) { c->u[(py / 2 + (i >> 1)) * c->sc + px / 2 + (i & 1)] = (uint8_t)ou[i];
            c->v[(py / 2 + (i >> 1)) * c->sc + px / 2 + (i & 1)] = (uint8_t)ov[i]; }
    }
}
static int sad_mb(Enc *e, int mx, int my) {
    const Frame *s = &e->src, *c = &e->cur; int sad = 0;
    for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) sad += ABS(s->y[(my * 16 + y) * s->sy + mx * 16 + x] - c->y[(my * 16 + y) * c->sy + mx * 16 + x]);
    return sad;
}
/* B-slice syntax elements (Tables 7-14, 7-18, 9-37) */
static void se_skip_flag_b(Enc *e, int mx, int my, int skip) {
    MbE *a = mb_avail(e, mx - 1, my), *b = mb_avail(e, mx, my - 1);
    cab_enc(&e->cab, 24 + (a && !a->skip) + (b && !b->skip), skip);
}
static void se_mb_type_b(Enc *e, int mx, int my, int t) {            /* t: 0..22 inter types */
    if (!e->cabac) { bw_ue(&e->bw, t); return; }
    CabEnc *c = &e->cab; MbE *a = mb_avail(e, mx - 1, my), *b = mb_avail(e, mx, my - 1);
    cab_enc(c, 27 + (a && !a->bdirect16) + (b && !b->bdirect16), t != 0);
    if (t == 0) return;
    if (t <= 2) { cab_enc(c, 30, 0); cab_enc(c, 32, t == 2); return; }
    cab_enc(c, 30, 1);
    int bits, extra = -1;                                                /* 4 bins, then one more for types 12..21 */
    if (t <= 10) bits = t - 3; else if (t == 11) bits = 14; else if (t == 22) bits = 15; else { bits = (t + 4) >> 1; extra = (t + 4) & 1; }
    cab_enc(c, 31, (bits >> 3) & 1); cab_enc(c, 32, (bits >> 2) & 1); cab_enc(c, 32, (bits >> 1) & 1); cab_enc(c, 32, bits & 1);
    if (extra >= 0) cab_enc(c, 32, extra);
}
static void se_mb_type_b_intra_prefix(Enc *e, int mx, int my) {        /* prefix 1 1 1 1 0 1 (bits == 13) before the intra suffix */
    CabEnc *c = &e->cab; MbE *a = mb_avail(e, mx - 1, my), *b = mb_avail(e, mx, my - 1);
    cab_enc(c, 27 + (a && !a->bdirect16) + (b && !b->bdirect16), 1);
    cab_enc(c, 30, 1); cab_enc(c, 31, 1); cab_enc(c, 32, 1); cab_enc(c, 32, 0); cab_enc(c, 32, 1);
}
static void se_sub_mb_type_b(Enc *e, int st) {
    if (!e->cabac) { bw_ue(&e->bw, st); return; }
    CabEnc *c = &e->cab;
    if (st == 0) { cab_enc(c, 36, 0); return; }
    cab_enc(c, 36, 1);
    if (st <= 2) { cab_enc(c, 37, 0); cab_enc(c, 39, st == 2); return; }
    cab_enc(c, 37, 1);
    if (st >= 11) { cab_enc(c, 38, 1); cab_enc(c, 39, 1); cab_enc(c, 39, st == 12); return; }
    if (st >= 7) { cab_enc(c, 38, 1); cab_enc(c, 39, 0); st -= 4; } else cab_enc(c, 38, 0);
    cab_enc(c, 39, ((st - 3) >> 1) & 1); cab_enc(c, 39, (st - 3) & 1);
}
static void se_ref_idx_l(Enc *e, int mx, int my, MbE *m, int l, int bx, int by, int nref, int v) {
    if (!e->cabac) { bw_te(&e->bw, nref - 1, v); return; }
    int inc = 0;
    for (int k = 0; k < 2; k++) { int r; MbE *n = nb4(e, mx, my, m, bx, by, k == 0, &r); int q = (r >> 3) * 2 + ((r & 3) >> 1);
        if (n && !n->intra && !(n->direct8 & (1 << q)) && mb_ref(n, l)[q] > 0) inc += k == 0 ? 1 : 2; }
    int ctx = 54 + inc;
    for (int i = 0; i < v; i++) { cab_enc(&e->cab, ctx, 1); ctx = 54 + (i == 0 ? 4 : 5); }
    cab_enc(&e->cab, ctx, 0);
}
static void se_mvd_l(Enc *e, int mx, int my, MbE *m, int l, int bx, int by, int bw, int bh, int dx, int dy) {
    if (!e->cabac) { bw_se(&e->bw, dx); bw_se(&e->bw, dy); return; }
    CabEnc *c = &e->cab;
    for (int comp = 0; comp < 2; comp++) {
        int d = comp ? dy : dx, a = ABS(d), sum = 0, base = comp ? 47 : 40;
        for (int k = 0; k < 2; k++) { int r; MbE *n = nb4(e, mx, my, m, bx, by, k == 0, &r); if (n && !n->intra) sum += (l ? n->mvd1 : n->mvd)[r][comp]; }
        cab_enc(c, base + (sum < 3 ? 0 : sum > 32 ? 2 : 1), a > 0);
        if (!a) continue;
        int v = 1, ctx = 3;
        while (v < 9) { int bin = a > v; cab_enc(c, base + ctx, bin); if (!bin) break; v++; if (ctx < 6) ctx++; }
        if (a >= 9) cab_ueg(c, a - 9, 3);
        cab_byp(c, d < 0);
    }
    for (int y = by; y < by + bh; y++) for (int x = bx; x < bx + bw; x++) { (l ? m->mvd1 : m->mvd)[y * 4 + x][0] = (uint8_t)MIN(ABS(dx), 255);
        (l ? m->mvd1 : m->mvd)[y * 4 + x][1] = (uint8_t)MIN(ABS(dy), 255); }
}

typedef struct { int bx, by, bw, bh, pred /*0 L0 1 L1 2 Bi 3 direct*/, shape, part, q; } BPart;

static void encode_b_mb(Enc *e, int mx, int my, MbE *m, int *skip_run) {
    static const uint8_t pair[9][2] = {{0, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 2}, {1, 2}, {2, 0}, {2, 1}, {2, 2}};
    static const uint8_t sub_pred[13] = {3, 0, 1, 2, 0, 0, 1, 1, 2, 2, 0, 1, 2}, sub_shape[13] = {0, 0, 0, 0, 1, 2, 1, 2, 1, 2, 3, 3, 3};
    int fuzz = e->p.mode == 1, px = mx * 16, py = my * 16;
    int nref[2] = { e->nlist0, e->nlist1 };
    int direct_ok = e->p.direct_temporal ? temporal_direct_ok(e, mx, my) : 1;
    static __thread MbCode mc; memset(&mc, 0, sizeof mc);
    /* ---- choose the macroblock type ---- */
    int mbt, sub[4] = {0, 0, 0, 0};
    if (fuzz) {
        int k = rnd_n(&e->rng, 24);
        if (k == 0 && !e->p.no_intra) { se_begin_mb(e, mx, my, skip_run); encode_intra_mb(e, mx, my, m, rnd_n(&e->rng, 12) == 0 ? 7 : -1); return; }
        if (k == 0) k = 6;
        if (k < 6) mbt = 0; else if (k < 11) mbt = 1 + rnd_n(&e->rng, 3); else if (k < 17) mbt = 4 + rnd_n(&e->rng, 18); else mbt = 22;
        if (mbt == 22) for (int i = 0; i < 4; i++) sub[i] = rnd_n(&e->rng, 13);
        if (!direct_ok) { if (mbt == 0) mbt = 3; for (int i = 0; i < 4; i++) if (sub[i] == 0) sub[i] = 3; }
    } else mbt = -1;                                                     /* decided below by SAD */
    BPart parts[16]; int np = 0, dmask = 0;
    int want[2][16][2]; memset(want, 0, sizeof want);                    /* real mode: vectors found by the search, per 4x4 */
    if (!fuzz) {
        int best = 1 << 30, cand = 0, mvp[2], r16[2][2] = {{0, 0}, {0, 0}}, cost[4];
        for (int l = 0; l < 2; l++) {                                   /* 16x16 search per list around that list's predictor */
            e->decoded_mask = 0; pred_mv_l(e, mx, my, m, l, 0, 0, 4, 0, 0, 0, mvp);
            MvRes r = search_block(e, l ? e->list1[0] : e->list0[0], px, py, 16, 16, mvp, mvp[0], mvp[1], e->p.search);
            r16[l][0] = r.mvx; r16[l][1] = r.mvy; cost[1 + l] = r.cost;
        }
        mb_init(e, m);
        if (direct_ok) { b_direct(e, mx, my, m, 15); mc_mb(e, mx, my, m); cost[0] = sad_mb(e, mx, my) - 64; } else cost[0] = 1 << 30;
        mb_init(e, m);
        for (int i = 0; i < 4; i++) { m->ref[i] = 0; m->ref1[i] = 0; }
        for (int r = 0; r < 16; r++) { m->mv[r][0] = (int16_t)r16[0][0]; m->mv[r][1] = (int16_t)r16[0][1]; m->mv1[r][0] = (int16_t)r16[1][0];
            m->mv1[r][1] = (int16_t)r16[1][1]; }
        mc_mb(e, mx, my, m); cost[3] = sad_mb(e, mx, my) + 32;
        mb_init(e, m);
        for (int i = 0; i < 4; i++) if (cost[i] < best) { best = cost[i]; cand = i; }
        mbt = cand;
        for (int l = 0; l < 2; l++) for (int r = 0; r < 16; r++) { want[l][r][0] = r16[l][0]; want[l][r][1] = r16[l][1]; }
        if (best > 16 * 16 * 12) { se_begin_mb(e, mx, my, skip_run); encode_intra_mb(e, mx, my, m, -1); return; }
    }
    /* ---- partitions in syntax order ---- */
    if (mbt == 0) { dmask = 15; m->bdirect16 = 1; }
    else if (mbt <= 3) { BPart p = {0, 0, 4, 4, mbt - 1, 0, 0, 0}; parts[np++] = p; }
    else if (mbt <= 21) { int sh = (mbt & 1) ? 2 : 1; for (int i = 0; i < 2; i++) { BPart p = {
        sh == 2 ? i * 2 : 0, sh == 1 ? i * 2 : 0, sh == 2 ? 2 : 4, sh == 1 ? 2 : 4, pair[(mbt - 4) >> 1][i], sh, i, 0 }; parts[np++] = p; } }
    else for (int q = 0; q < 4; q++) {
        int pr = sub_pred[sub[q]], sp = sub_shape[sub[q]], ox = (q & 1) * 2, oy = (q >> 1) * 2;
        if (pr == 3) { dmask |= 1 << q; BPart p = {ox, oy, 2, 2, 3, 0, 0, q}; parts[np++] = p; continue; }
        int n = sp == 0 ? 1 : sp == 3 ? 4 : 2, bw = (sp == 0 || sp == 1) ? 2 : 1, bh = (sp == 0 || sp == 2) ? 2 : 1;
        for (int i = 0; i < n; i++) { BPart p = { ox + (sp == 1 ? 0 : sp == 2 ? i : (i & 1)), oy + (sp == 1 ? i : sp == 2 ? 0 : (i >> 1)), bw, bh, pr, 0, 0,
            q }; parts[np++] = p; }
    }
    if (dmask) b_direct(e, mx, my, m, dmask);
    /* reference indices */
    int refs[2][16];
    for (int l = 0; l < 2; l++) for (int i = 0; i < np; i++) {
        BPart *p = &parts[i]; refs[l][i] = -1;
        if (p->pred == 3 || !(p->pred == 2 || p->pred == l)) continue;
        if (mbt == 22 && i > 0 && parts[i - 1].q == p->q && parts[i - 1].pred != 3) refs[l][i] = refs[l][i - 1];   /* one ref_idx per sub-macroblock */
        else refs[l][i] = (fuzz && nref[l] > 1) ? rnd_n(&e->rng, nref[l]) : 0;
        for (int y = p->by; y < p->by + p->bh; y++) for (int x = p->bx; x < p->bx + p->bw; x++) mb_ref(m, l)[(y >> 1) * 2 + (x >> 1)] = (int8_t)refs[l][i];
    }
    /* motion vectors, list by list in syntax order */
    int mvd[2][16][2]; memset(mvd, 0, sizeof mvd);
    for (int l = 0; l < 2; l++) {
        e->decoded_mask = 0;
        for (int i = 0; i < np; i++) {
            BPart *p = &parts[i];
            if (refs[l][i] < 0) { for (int y = p->by; y < p->by + p->bh; y++) for (int x = p->bx; x < p->bx + p->bw; x++) e->decoded_mask |= 1 << (y * 4 + x);
                continue; }
            int mvp[2], mv[2];
            pred_mv_l(e, mx, my, m, l, p->bx, p->by, p->bw, refs[l][i], p->shape, p->part, mvp);
            if (fuzz) random_mv(e, px + p->bx * 4, py + p->by * 4, p->bw * 4, p->bh * 4, mvp, mv);
            else { mv[0] = want[l][p->by * 4 + p->bx][0]; mv[1] = want[l][p->by * 4 + p->bx][1]; }
            mvd[l][i][0] = mv[0] - mvp[0]; mvd[l][i][1] = mv[1] - mvp[1];
            store_mv_l(e, m, l, p->bx, p->by, p->bw, p->bh, mv[0], mv[1]);
        }
    }
    for (int q = 0; q < 4; q++) { m->refid[q] = m->ref[q] >= 0 ? e->list0[m->ref[q]]->id : -1; m->refid1[q] = m->ref1[q] >= 0 ? e->list1[m->ref1[q]]->id : -1; }
    /* ---- prediction + residual ---- */
    mc_mb(e, mx, my, m);
    int try_skip = mbt == 0 && (fuzz ? rnd_n(&e->rng, 3) != 0 : 0);
    int dqp = 0;
    if (fuzz && rnd_n(&e->rng, 6) == 0) dqp = rnd_n(&e->rng, 9) - 4;
    int qp = CLIP3(10, 48, e->qp_run + dqp); dqp = qp - e->qp_run;
    int small = 0;                                                       /* a partition below 8x8, or direct without direct_8x8_inference */
    for (int i = 0; i < np; i++) if (parts[i].bw < 2 || parts[i].bh < 2 || (parts[i].pred == 3 && !e->p.dinf8)) small = 1;
    if (mbt == 0 && !e->p.dinf8) small = 1;
    int t8_ok = e->p.t8x8 && !small, use_t8 = t8_ok && (fuzz ? rnd_n(&e->rng, 2) : 1), cbp_l = 0;
    if (!try_skip && use_t8) { for (int b8 = 0; b8 < 4; b8++) if (code_luma8(e, px + (b8 & 1) * 8, py + (b8 >> 1) * 8, qp, 0, mc.luma8[b8])) cbp_l |= 1 << b8; }
    else if (!try_skip) for (int blk = 0; blk < 16; blk++) { int bx = bX(blk), by = bY(blk);
        if (code_luma4(e, px + bx * 4, py + by * 4, qp, 0, mc.luma[by * 4 + bx])) cbp_l |= 1 << (blk >> 2); }
    if (!cbp_l) use_t8 = 0;
    mc.t8 = use_t8;
    int qpc = chroma_qp_of(e, qp), cflags = 0;
    if (!try_skip) for (int pl = 0; pl < 2; pl++) cflags |= code_chroma(e, mx, my, pl, qpc, 0, &mc);
    int cbp_c = (cflags & 2) ? 2 : (cflags & 1) ? 1 : 0;
    for (int pl = 0; pl < 2; pl++) recon_chroma(e, mx, my, pl, qpc, &mc, cbp_c >= 1, cbp_c == 2);
    mc.cbp = cbp_l | (cbp_c << 4); mc.type = 0;
    /* ---- B_Skip ---- */
    if (mbt == 0 && mc.cbp == 0 && !(fuzz && rnd_n(&e->rng, 4) == 0)) {
        m->skip = 1; m->qp = (uint8_t)e->qp_run; m->qpc[0] = m->qpc[1] = (uint8_t)chroma_qp_of(e, e->qp_run);
        if (e->cabac) { se_skip_flag_b(e, mx, my, 1); e->last_dqp = 0; } else (*skip_run)++;
        return;
    }
    if (e->cabac) se_skip_flag_b(e, mx, my, 0); else { bw_ue(&e->bw, *skip_run); *skip_run = 0; }
    se_mb_type_b(e, mx, my, mbt);
    if (mbt == 22) for (int q = 0; q < 4; q++) se_sub_mb_type_b(e, sub[q]);
    for (int l = 0; l < 2; l++) for (int i = 0; i < np; i++) {
        BPart *p = &parts[i];
        if (refs[l][i] < 0 || nref[l] <= 1) continue;
        if (mbt == 22 && i > 0 && parts[i - 1].q == p->q) continue;       /* ref_idx once per sub-macroblock */
        se_ref_idx_l(e, mx, my, m, l, mbt == 22 ? (p->q & 1) * 2 : p->bx, mbt == 22 ? (p->q >> 1) * 2 : p->by, nref[l], refs[l][i]);
    }
    for (int l = 0; l < 2; l++) for (int i = 0; i < np; i++) if (refs[l][i] >= 0) se_mvd_l(e, mx, my, m, l, parts[i].bx, parts[i].by, parts[i].bw, parts[i].bh,
        mvd[l][i][0], mvd[l][i][1]);
    se_cbp(e, mx, my, mc.cbp, 0); m->cbp = (uint8_t)mc.cbp;
    if (cbp_l && t8_ok) se_t8_flag(e, mx, my, use_t8);
    m->t8 = (uint8_t)use_t8;
    if (mc.cbp > 0) { se_dqp(e, dqp); e->qp_run = qp; } else { qp = e->qp_run; e->last_dqp = 0; }
    m->qp = (uint8_t)qp; m->qpc[0] = m->qpc[1] = (uint8_t)chroma_qp_of(e, qp);
    if (mc.cbp > 0) write_mb_residual(e, mx, my, m, &mc);
}

/* ------------------------------ headers -------------------------------------- */
static const uint8_t dflt4_intra[16] = {6,13,13,20,20,20,28,28,28,28,32,32,32,37,37,42}, dflt4_inter[16] = {10,14,14,20,20,20,24,24,24,24,27,27,27,30,30,34};
static const uint8_t dflt8_intra[64] = {6,10,10,13,11,13,16,16,16,16,18,18,18,18,18,23,23,23,23,23,23,25,25,25,25,25,25,25,27,27,27,27,
    27,27,27,27,29,29,29,29,29,29,29,31,31,31,31,31,31,33,33,33,33,33,36,36,36,36,38,38,38,40,40,42};
static const uint8_t dflt8_inter[64] = {9,13,13,15,13,15,17,17,17,17,19,19,19,19,19,21,21,21,21,21,21,22,22,22,22,22,22,22,24,24,24,24,
    24,24,24,24,25,25,25,25,25,25,25,27,27,27,27,27,27,28,28,28,28,28,30,30,30,30,32,32,32,33,33,35};
/* scaling_matrix syntax (7.3.2.1.1 / 7.3.2.2) with fall-back rule A; chooses per list: absent, use-default, full list, truncated list.
 * Leaves the effective matrices in g_w4 / g_w8 (raster order). */
static void write_scaling_matrix(Enc *e, BitW *w, int n_lists) {
    uint8_t eff[8][64]; Rng r; r.s = (uint64_t)e->p.seed * 0x51ED27ull + 99;      /* own generator: identical matrices in every SPS / PPS of the stream */
    for (int i = 0; i < 8; i++) {
        int n = i < 6 ? 16 : 64;
        const uint8_t *dflt = i < 6 ? (i < 3 ? dflt4_intra : dflt4_inter) : (i == 6 ? dflt8_intra : dflt8_inter);
        int choice = rnd_n(&r, 4);
        if (i >= n_lists) { memcpy(eff[i], dflt, n); memset(eff[i], 16, n); continue; }   /* 8x8 lists not sent: flat (transform_8x8_mode off) */
        if (choice == 0) {                                                 /* not present: fall-back rule A */
            bw_put(w, 1, 0);
            if (i == 0 || i == 3 || i >= 6) memcpy(eff[i], dflt, n); else memcpy(eff[i], eff[i - 1], n);
            continue;
        }
        bw_put(w, 1, 1);
        if (choice == 1) { bw_se(w, -8); memcpy(eff[i], dflt, n); continue; }          /* useDefaultScalingMatrixFlag */
        int last = 8, stop = choice == 3 ? 1 + rnd_n(&r, n - 1) : n;
        for (int j = 0; j < n; j++) {
            if (j == stop) { bw_se(w, -last <= -128 ? 256 - last : -last); for (int k = j; k < n; k++) eff[i][k] = (uint8_t)last; break; }
            int v = 8 + rnd_n(&r, 41);                                     /* 8..48 */
            if (j > 0 && rnd_n(&r, 3)) { const int step = last + rnd_n(&r, 9) - 4; v = CLIP3(8, 48, step); }
                /* (CLIP3 is a macro: the draw must not sit inside it -- it used to,
                                                                                                        and one matrix in ~150 came out with a 0 entry,
                                                                                                            which ends the list early:
                                                                                                            both decoders rightly rejected that SPS) */
            int d = v - last; bw_se(w, d);
            eff[i][j] = (uint8_t)v; last = v;
        }
    }
    for (int i = 0; i < 6; i++) for (int k = 0; k < 16; k++) g_w4[i][zz4[k]] = eff[i][k];
    for (int i = 0; i < 2; i++) for (int k = 0; k < 64; k++) g_w8[i][zz8[k]] = eff[6 + i][k];
    g_w8_version++;
}
static void write_sps_pps(Enc *e) {
    BitW *w = &e->bw; GenParams *p = &e->p;
    w->len = 0; w->nbits = 0; w->cur = 0;
    for (int i = 0; i < 6; i++) for (int k = 0; k < 16; k++) g_w4[i][k] = 16;
    for (int i = 0; i < 2; i++) for (int k = 0; k < 64; k++) g_w8[i][k] = 16;
    g_w8_version++;
    int high = p->t8x8 || p->scaling;
    if (high) { bw_put(w, 8, 100); bw_put(w, 8, 0); }                  /* High */
    else if (p->cabac || p->bframes || p->wp || p->fmo0) { bw_put(w, 8, 77); bw_put(w, 8, 0x40); }          /* Main, constraint_set1 */
    else { bw_put(w, 8, 66); bw_put(w, 8, 0xC0); }                        /* Baseline, constraint_set0/1 */
    bw_put(w, 8, p->level_idc);
    bw_ue(w, 0);
    if (high) { bw_ue(w, 1); bw_ue(w, 0); bw_ue(w, 0); bw_put(w, 1, 0); bw_put(w, 1, (uint32_t)(p->scaling == 1));
        if (p->scaling == 1) write_scaling_matrix(e, w, 8); }   /* 4:2:0, 8 bit, no bypass, scaling matrix */
    bw_ue(w, e->log2_max_fn - 4);
    bw_ue(w, p->poc_type);
    if (p->poc_type == 0) bw_ue(w, e->poc_lsb_bits - 4);
    if (p->poc_type == 1) {                                               /* delta_pic_order_always_zero_flag 0, offsets, the cycle */
        bw_put(w, 1, 0); bw_se(w, e->t1_nonref); bw_se(w, e->t1_t2b); bw_ue(w, e->t1_cycle);
        for (int i = 0; i < e->t1_cycle; i++) bw_se(w, e->t1_ref[i]);
    }
    bw_ue(w, p->num_ref); bw_put(w, 1, (uint32_t)(p->gaps != 0));      /* max_num_ref_frames, gaps_in_frame_num_value_allowed_flag */
    bw_ue(w, e->mbw - 1); bw_ue(w, (p->fmo0 ? e->mbh / 2 : e->mbh) - 1);   /* pic_height_in_map_units: field macroblock rows when frame_mbs_only_flag = 0 */
    if (p->fmo0) { bw_put(w, 1, 0); bw_put(w, 1, 0); bw_put(w, 1, 1); }  /* frame_mbs_only_flag 0, mb_adaptive_frame_field_flag 0,
        direct_8x8_inference_flag 1 */
    else { bw_put(w, 1, 1); bw_put(w, 1, (uint32_t)p->dinf8); }           /* frame_mbs_only, direct_8x8_inference */
    int cr = (e->W - p->width) / 2, cb = (e->H - p->height) / (p->fmo0 ? 4 : 2);   /* CropUnitY = 2 * (2 - frame_mbs_only_flag) */
    if (cr || cb) { bw_put(w, 1, 1); bw_ue(w, 0); bw_ue(w, cr); bw_ue(w, 0); bw_ue(w, cb); } else bw_put(w, 1, 0);
    if (p->vui_fps > 0) {                                                 /* vui_parameters() (E.1.1) */
        bw_put(w, 1, 1);
        bw_put(w, 1, 1); bw_put(w, 8, 1);                                 /* aspect_ratio_info_present_flag, aspect_ratio_idc 1 (square) */
        bw_put(w, 1, 0); bw_put(w, 1, 0); bw_put(w, 1, 0);                /* no overscan info, video signal type, chroma location */
        bw_put(w, 1, 1); bw_put(w, 32, 1); bw_put(w, 32, 2u * (uint32_t)p->vui_fps); bw_put(w, 1, 1);   /* timing_info: tick, time_scale, fixed rate */
        bw_put(w, 1, 0); bw_put(w, 1, 0);                                 /* no NAL / VCL HRD parameters */
        bw_put(w, 1, 0); bw_put(w, 1, 0);                                 /* pic_struct_present_flag, bitstream_restriction_flag */
    } else bw_put(w, 1, 0);                                               /* no VUI */
    bw_trailing(w); out_nal(&e->out, 3, 7, w, 1);
    w->len = 0;
    bw_ue(w, 0); bw_ue(w, 0); bw_put(w, 1, (uint32_t)(p->cabac != 0)); bw_put(w, 1, (uint32_t)p->poc_bottom); bw_ue(w, 0);
    /* ..., bottom_field_pic_order_in_frame_present_flag, one slice group */
    bw_ue(w, p->num_ref - 1); bw_ue(w, 0);
    bw_put(w, 1, (uint32_t)(p->wp == 1)); bw_put(w, 2, (uint32_t)(p->bframes ? p->wp : 0));   /* weighted_pred_flag, weighted_bipred_idc */
    bw_se(w, p->qp - 26); bw_se(w, 0); bw_se(w, p->chroma_qp_off);
    bw_put(w, 1, 1); bw_put(w, 1, p->cip); bw_put(w, 1, (uint32_t)(p->redundant != 0));      /* deblocking_filter_control_present, constrained_intra_pred,
        redundant_pic_cnt_present */
    if (high) { bw_put(w, 1, (uint32_t)p->t8x8); bw_put(w, 1, (uint32_t)(p->scaling == 2)); if (p->scaling == 2) write_scaling_matrix(e, w, 6 + 2 * p->t8x8);
        bw_se(w, p->chroma_qp_off); }   /* transform_8x8_mode, scaling matrix, second_chroma_qp_index_offset */
    bw_trailing(w); out_nal(&e->out, 3, 8, w, 1);
}

/* copy the lines of one parity of a frame into a field picture / back (the field pictures are pictures of half the height) */
static void field_copy(Frame *frm, Frame *fld, int par, int W, int H, int to_field) {
    for (int y = 0; y < H / 2; y++) {
        uint8_t *a = frm->y + (2 * y + par) * frm->sy, *b = fld->y + y * fld->sy;
        if (to_field) memcpy(b, a, W); else memcpy(a, b, W);
    }
    for (int pl = 0; pl < 2; pl++) for (int y = 0; y < H / 4; y++) {
        uint8_t *a = (pl ? frm->v : frm->u) + (2 * y + par) * frm->sc, *b = (pl ? fld->v : fld->u) + y * fld->sc;
        if (to_field) memcpy(b, a, W / 2); else memcpy(a, b, W / 2);
    }
}
/* 8.2.4.2.5: the fields of an ordered list of frame stores, alternating in parity and starting with the parity of the current field; a store whose
   field of the wanted parity is not marked `mark` is passed over, and when one parity runs out the rest of the other follows in order */
static int alternate_fields(Frame **stores, int n, int mark, int par, Frame **out, int cnt) {
    int c[2] = {0, 0}, q = par;
    for (;;) {
        while (c[q] < n && stores[c[q]]->fmark[q] != mark) c[q]++;
        if (c[q] < n) { out[cnt++] = &stores[c[q]]->fld[q]; c[q]++; }
        else { int o = q ^ 1; while (c[o] < n && stores[c[o]]->fmark[o] != mark) c[o]++; if (c[o] >= n) break; }
        q ^= 1;
    }
    return cnt;
}

/* t = display index; is_b: a B picture (non-reference, coded after its following anchor), otherwise P (or IDR / I when t starts a GOP);
   field: 0 = a frame picture, 1 / 2 = the top / bottom field of the frame as a picture of its own; second: it is the second field of its frame */
static void encode_picture(Enc *e, int t, int is_b, int field, int second) {
    GenParams *p = &e->p; BitW *w = &e->bw;
    /* an IDR frame coded as two fields: the first field is the IDR picture, the second a P (or I) field that can only see the first (7.4.1.2.4) */
    int idr = !is_b && (t % p->gop) == 0 && !second, after_idr = second && (t % p->gop) == 0;
    int is_ref = !is_b && (idr || after_idr || !(p->nonref_period > 0 && (t % p->gop) % p->nonref_period == p->nonref_period - 1 && (t + 1) % p->gop != 0));
    if (p->bframes) is_ref = !is_b;
    e->field = field; e->second = second;
    e->scan4 = field ? fs4 : zz4; e->scan8 = field ? fs8 : zz8;
    const int par = field ? field - 1 : 0;
    if (!second) render_source(e, t);
    if (idr) { e->frame_num = 0; e->nrefs = 0; e->poc_base = t; write_sps_pps(e); }
    /* (not right after operation 5: that picture's frame_num counts as 0 afterwards, the next one has 1 -- with a gap it could repeat the very frame_num
       and order count the operation-5 picture was sent with, and 7.4.1.2.4 could not tell the two apart) */
    const int gap_ok = !e->after_op5;
    if (!second) e->after_op5 = 0;
    if (p->gaps && !idr && !is_b && !second && gap_ok && e->nrefs > 0 && rnd_n(&e->rng, 4) == 0) {
        /* 8.2.5.2: one or two frame_num values are left out.  For each of them the decoder infers a frame: the sliding window runs as for any
           reference frame, and the frame stays in the buffer as a short-term reference without samples */
        const int mfn = 1 << e->log2_max_fn;
        for (int k = 1 + rnd_n(&e->rng, 2); k > 0; k--) {
            const int fn = e->frame_num & (mfn - 1);
            if (e->nrefs >= p->num_ref) {
                int old = -1;
#define WRAP_(f) ((f)->frame_num > fn ? (f)->frame_num - mfn : (f)->frame_num)
                for (int i = 0; i < e->nrefs; i++) if ((p->paff ? e->refs[i].fmark[0] == 1 || e->refs[i].fmark[1] == 1 : !e->refs[i].is_long) &&
                    (old < 0 || WRAP_(&e->refs[i]) < WRAP_(&e->refs[old]))) old = i;
#undef WRAP_
                if (old < 0) break;                                        /* only long-term pictures left: the window has nothing to drop */
                { Frame t_ = e->refs[old]; for (int q_ = old; q_ + 1 < e->nrefs; q_++) e->refs[q_] = e->refs[q_ + 1]; e->nrefs--; e->refs[e->nrefs] = t_; }
            }
            Frame *f = &e->refs[e->nrefs++];
            f->frame_num = fn; f->is_long = 0; f->lt_idx = -1; f->nonexist = 1; f->fmark[0] = f->fmark[1] = 1; f->coded_fields = 0; f->id = e->next_id++;
            f->poc = f->fpoc[0] = f->fpoc[1] = 0;
            e->frame_num++;
        }
    }
    /* the store of the frame being coded: a frame, or the first field of a non-reference frame, is coded into e->cur; the second field of a reference
       frame into the store its first field already opened (e->refs[cur_store]) */
    int cur_store = -1;
    if (second && is_ref) { for (int i = 0; i < e->nrefs; i++) if (e->refs[i].id == e->cur_store_id) cur_store = i;
        if (cur_store < 0) { fprintf(stderr, "h264gen: the first field's store is gone\n"); abort(); } }
    Frame full_src = e->src, spare_cur = e->cur, full_cur = cur_store >= 0 ? e->refs[cur_store] : e->cur;
    if (field) {
        /* from here to the end of the picture the encoder sees a picture of half the height */
        field_copy(&full_src, &e->fsrc, par, e->W, e->H, 1);
        e->src = e->fsrc; e->cur = full_cur.fld[par]; e->cur.parity = par;
        e->H /= 2; e->mbh /= 2;
    }
    e->slice_type = idr ? 2 : (is_b ? 1 : 0);
    if (!idr && !is_b && p->mode == 1 && rnd_n(&e->rng, 12) == 0 && !p->no_intra) e->slice_type = 2;   /* occasional non-IDR I picture */
    /* 8.2.1: TopFieldOrderCnt counts 2 per picture from the last restart; with poc_bottom the bottom field may lie one below or above, and
       PicOrderCnt(frame) = Min(top, bottom) is what the lists, direct prediction and the output order see.  Of two field pictures the first has the
       frame's count, the second lies one above */
    if (!second) {
        e->cur_top = 2 * (t - e->poc_base); e->delta_bottom = 0; e->delta0 = 0;
        if (p->poc_bottom && !field) { e->delta_bottom = rnd_n(&e->rng, 3) - 1; if (idr && e->delta_bottom < 0) e->delta_bottom = 1; }
            /* an IDR frame: Min(top, bottom) = 0 (8.2.1) */
        if (field) { e->delta_bottom = field == 1 ? 1 : -1; if (field == 2) e->cur_top += 1; }      /* bottom first: top = count + 1, bottom = count */
        e->cur_poc = e->cur_top + MIN(0, e->delta_bottom);
        if (e->pocs) e->pocs[t] = e->cur_poc;
    }
    if (p->poc_type == 1) e->delta0 = rnd_n(&e->rng, 2);
    const int field_poc = field == 2 ? e->cur_top + e->delta_bottom : e->cur_top;       /* the count this picture's slice headers carry (type 0) */
    if (field) e->cur_poc = field_poc;          /* what direct prediction and implicit weights measure distances from (8.4.1.2.3, 8.4.2.3.1) */
    e->cur.poc = e->cur_poc;
    const int maxfn = 1 << e->log2_max_fn, curfn = e->frame_num & (maxfn - 1);
    e->cur.frame_num = curfn; e->cur.is_long = 0; e->cur.lt_idx = -1;
#define PICNUM(f) ((f)->frame_num > curfn ? (f)->frame_num - maxfn : (f)->frame_num)
    /* 8.2.4.1: picture numbers.  In a field picture every reference FIELD has one: twice the frame's number, plus one for a field of the current parity */
#define PN(f) (field ? 2 * PICNUM(f) + ((f)->parity == par) : PICNUM(f))
#define LPN(f) (field ? 2 * (f)->lt_idx + ((f)->parity == par) : (f)->lt_idx)
    const int cur_pn = field ? 2 * curfn + 1 : curfn, max_pn = field ? 2 * maxfn : maxfn;
    Frame *init[2][16]; int ninit[2] = {0, 0};
    if (field && is_b) {
        /* 8.2.4.2.4 + 8.2.4.2.5 (B field): frame stores by PicOrderCnt around the current FIELD's count -- list 0: those not above it, descending,
           then the others ascending; list 1 the other way round --, long-term stores behind; from each ordered list the fields, alternating in
           parity.  (All pictures of a stream with B pictures are field pairs here, so every store has both fields marked.) */
        Frame *before[6], *after[6], *ord[2][6]; int nb = 0, na = 0;
        for (int i = 0; i < e->nrefs; i++) {
            Frame *s = &e->refs[i];
            for (int q = 0; q < 2; q++) { Frame *f = &s->fld[q]; f->frame_num = s->frame_num; f->is_long = s->fmark[q] == 2; f->lt_idx = s->lt_idx; f->parity = q;
                f->poc = s->fpoc[q]; f->id = (1 << 20) + 2 * s->id + q; f->store_mf = s->mf; f->store_fields = s->coded_fields; f->store_id = s->id; f->nonexist = s->nonexist; }
            if (s->fmark[0] != 1 && s->fmark[1] != 1) continue;
            if (s->poc <= e->cur_poc) before[nb++] = s; else after[na++] = s;
        }
        for (int i = 0; i < nb; i++) for (int j = i + 1; j < nb; j++) if (before[j]->poc > before[i]->poc) { Frame *x = before[i]; before[i] = before[j];
            before[j] = x; }
        for (int i = 0; i < na; i++) for (int j = i + 1; j < na; j++) if (after[j]->poc < after[i]->poc) { Frame *x = after[i]; after[i] = after[j];
            after[j] = x; }
        for (int i = 0; i < nb; i++) { ord[0][i] = before[i]; ord[1][na + i] = before[i]; }
        for (int i = 0; i < na; i++) { ord[0][nb + i] = after[i]; ord[1][i] = after[i]; }
        for (int l = 0; l < 2; l++) ninit[l] = alternate_fields(ord[l], nb + na, 1, par, init[l], 0);
        if (ninit[1] > 1 && ninit[0] == ninit[1]) { int same = 1; for (int i = 0; i < ninit[0]; i++) if (init[0][i] != init[1][i]) same = 0;
            if (same) { Frame *x = init[1][0]; init[1][0] = init[1][1]; init[1][1] = x; } }
        e->nlist0 = MIN(ninit[0], 2 * p->num_ref); e->nlist1 = MIN(ninit[1], 2 * p->num_ref);
        if (p->mode == 1) { e->nlist0 = 1 + rnd_n(&e->rng, e->nlist0); e->nlist1 = 1 + rnd_n(&e->rng, e->nlist1); }   /* num_ref_idx_active override */
        for (int i = 0; i < e->nlist0; i++) e->list0[i] = init[0][i];
        for (int i = 0; i < e->nlist1; i++) e->list1[i] = init[1][i];
    } else if (field) {
        /* 8.2.4.2.2 + 8.2.4.2.5 (P field): frame stores with any field marked short-term by descending FrameNumWrap -- the store of the first field
           of this frame among them, at the front --, then those with a long-term field by ascending LongTermFrameIdx; from each list the fields */
        Frame *sh[6], *lg[6]; int ns = 0, nl = 0;
        for (int i = 0; i < e->nrefs; i++) {
            Frame *s = &e->refs[i];
            for (int q = 0; q < 2; q++) { Frame *f = &s->fld[q]; f->frame_num = s->frame_num; f->is_long = s->fmark[q] == 2; f->lt_idx = s->lt_idx; f->parity = q;
                f->poc = s->fpoc[q]; f->id = (1 << 20) + 2 * s->id + q; f->store_mf = s->mf; f->store_fields = s->coded_fields; f->store_id = s->id; f->nonexist = s->nonexist; }
            if (s->fmark[0] == 1 || s->fmark[1] == 1) sh[ns++] = s;
            if (s->fmark[0] == 2 || s->fmark[1] == 2) lg[nl++] = s;
        }
        for (int i = 0; i < ns; i++) for (int j = i + 1; j < ns; j++) if (PICNUM(sh[j]) > PICNUM(sh[i])) { Frame *x = sh[i]; sh[i] = sh[j]; sh[j] = x; }
        for (int i = 0; i < nl; i++) for (int j = i + 1; j < nl; j++) if (lg[j]->lt_idx < lg[i]->lt_idx) { Frame *x = lg[i]; lg[i] = lg[j]; lg[j] = x; }
        ninit[0] = alternate_fields(sh, ns, 1, par, init[0], 0);
        ninit[0] = alternate_fields(lg, nl, 2, par, init[0], ninit[0]);
        e->nlist0 = MIN(ninit[0], 2 * p->num_ref); e->nlist1 = 0;
        if (p->mode == 1 && e->nlist0 > 1 && rnd_n(&e->rng, 3) == 0) e->nlist0 = 1 + rnd_n(&e->rng, e->nlist0);      /* num_ref_idx_active override */
        for (int i = 0; i < e->nlist0; i++) e->list0[i] = init[0][i];
        if (e->nlist0 == 0 && e->slice_type == 0) e->slice_type = 2;
    } else if (!is_b) {
        /* 8.2.4.2.1: short-term references by descending PicNum (most recent first), then long-term by ascending LongTermPicNum */
        Frame *sh[5], *lg[5]; int ns = 0, nl = 0;
        for (int i = 0; i < e->nrefs; i++) {
            if (p->paff && e->refs[i].fmark[0] != e->refs[i].fmark[1]) continue;      /* a frame needs both of its fields to be a reference frame */
            if (e->refs[i].is_long) lg[nl++] = &e->refs[i]; else sh[ns++] = &e->refs[i];
        }
        if (ns + nl == 0 && e->slice_type == 0) e->slice_type = 2;
        for (int i = 0; i < ns; i++) for (int j = i + 1; j < ns; j++) if (PICNUM(sh[j]) > PICNUM(sh[i])) { Frame *x = sh[i]; sh[i] = sh[j]; sh[j] = x; }
        for (int i = 0; i < nl; i++) for (int j = i + 1; j < nl; j++) if (lg[j]->lt_idx < lg[i]->lt_idx) { Frame *x = lg[i]; lg[i] = lg[j]; lg[j] = x; }
        for (int i = 0; i < ns; i++) init[0][ninit[0]++] = sh[i];
        for (int i = 0; i < nl; i++) init[0][ninit[0]++] = lg[i];
        e->nlist0 = MIN(ninit[0], p->num_ref); e->nlist1 = 0;
        for (int i = 0; i < e->nlist0; i++) e->list0[i] = init[0][i];
    } else {
        /* 8.2.4.2.3: list 0 = earlier pictures by descending POC then later ones ascending; list 1 the other way round */
        Frame *before[5], *after[5]; int nb = 0, na = 0, n = MIN(e->nrefs, p->num_ref);
        for (int i = 0; i < n; i++) { if (e->refs[i].poc < e->cur_poc) before[nb++] = &e->refs[i]; else after[na++] = &e->refs[i]; }
        for (int i = 0; i < nb; i++) for (int j = i + 1; j < nb; j++) if (before[j]->poc > before[i]->poc) { Frame *x = before[i]; before[i] = before[j];
            before[j] = x; }
        for (int i = 0; i < na; i++) for (int j = i + 1; j < na; j++) if (after[j]->poc < after[i]->poc) { Frame *x = after[i]; after[i] = after[j];
            after[j] = x; }
        e->nlist0 = e->nlist1 = 0;
        for (int i = 0; i < nb; i++) e->list0[e->nlist0++] = before[i];
        for (int i = 0; i < na; i++) e->list0[e->nlist0++] = after[i];
        for (int i = 0; i < na; i++) e->list1[e->nlist1++] = after[i];
        for (int i = 0; i < nb; i++) e->list1[e->nlist1++] = before[i];
        if (e->nlist1 > 1) { int same = 1; for (int i = 0; i < e->nlist0; i++) if (e->list0[i] != e->list1[i]) same = 0; if (same) { Frame *x = e->list1[0];
            e->list1[0] = e->list1[1]; e->list1[1] = x; } }
        for (int i = 0; i < e->nlist0; i++) init[0][ninit[0]++] = e->list0[i];
        for (int i = 0; i < e->nlist1; i++) init[1][ninit[1]++] = e->list1[i];
        if (p->mode == 1) { e->nlist0 = 1 + rnd_n(&e->rng, e->nlist0); e->nlist1 = 1 + rnd_n(&e->rng, e->nlist1); }   /* num_ref_idx_active override */
    }
    /* 8.2.4.3: random modification of the initial lists (fuzz).  The final list is computed with the clause's own insertion rule */
    e->n_mod[0] = e->n_mod[1] = 0;
    if (p->rplm && e->slice_type != 2) for (int l = 0; l < (is_b ? 2 : 1); l++) {
        int nact = l ? e->nlist1 : e->nlist0;
        if (nact < 1 || rnd_n(&e->rng, 2)) continue;
        Frame *list[20]; for (int i = 0; i < 20; i++) list[i] = i < ninit[l] && i < nact ? init[l][i] : NULL;
        int m = 1 + rnd_n(&e->rng, MIN(nact, 3)), pred = cur_pn, idx = 0;
        for (int k = 0; k < m; k++) {
            Frame *tg = init[l][rnd_n(&e->rng, ninit[l])];
            if (tg->nonexist) { m = k; break; }                          /* (a frame that was never sent is not named) */
            if (tg->is_long) { e->mod_idc[l][k] = 2; e->mod_val[l][k] = LPN(tg); }
            else {
                int pn = PN(tg), nowrap = pn < 0 ? pn + max_pn : pn, diff = nowrap - pred;
                if (diff > 0) { e->mod_idc[l][k] = 1; e->mod_val[l][k] = diff - 1; }
                else if (diff < 0) { e->mod_idc[l][k] = 0; e->mod_val[l][k] = -diff - 1; }
                else { e->mod_idc[l][k] = 0; e->mod_val[l][k] = max_pn - 1; }           /* a full turn lands on the same PicNum */
                pred = nowrap;
            }
            for (int c = nact; c > idx; c--) list[c] = list[c - 1];
            list[idx++] = tg;
            int n = idx;
            for (int c = idx; c <= nact; c++) if (list[c] != tg) list[n++] = list[c];
        }
        e->n_mod[l] = m;
        int cnt = 0;
        for (int i = 0; i < nact; i++) if (list[i]) { (l ? e->list1 : e->list0)[cnt++] = list[i]; } else break;
        if (l) e->nlist1 = cnt; else e->nlist0 = cnt;
    }
    e->n_usable = 0;
    for (int i = 0; i < e->nlist0 && i < 16; i++) if (!(field ? e->list0[i]->nonexist : e->list0[i]->nonexist)) e->usable[e->n_usable++] = i;
    if (e->slice_type == 0 && e->n_usable == 0) e->slice_type = 2;      /* nothing to predict from */
    /* 8.2.5: marking of this picture (decided before the slices are written; applied after the picture is coded) */
    e->n_mmco = 0; e->idr_long = 0;
    int half_stores = 0;            /* PAFF: frame stores of which only one field is (still) a reference; no frame picture number names them */
    if (p->paff) for (int i = 0; i < e->nrefs; i++) half_stores += e->refs[i].fmark[0] != e->refs[i].fmark[1];
    if (p->mmco && is_ref && !is_b && field) {
        /* Field pictures (8.2.5.4 with field picture numbers): now and then one reference FIELD is dropped (operation 1), or every field of one frame
           store (operation 1 per short-term field, operation 2 per long-term field).  The first field of a frame needs a free frame store and, with
           operations present, gets no sliding window: stores are dropped until there is one.  Long-term fields: both fields of a store are turned
           long-term by two operations 3 with the same LongTermFrameIdx (the second must not free the index the first just took: 8.2.5.4.3), or the
           current field by operation 6 -- then the second field of the frame follows with the same index --, after operation 4 has allowed indices.
           (Operation 5 appears in the frame pictures of the stream only.) */
        uint8_t sim[5][2]; int n_used = 0, sim_lt[5], maxlt = e->max_lt_idx;
        for (int i = 0; i < e->nrefs; i++) { sim[i][0] = e->refs[i].fmark[0]; sim[i][1] = e->refs[i].fmark[1]; n_used += sim[i][0] || sim[i][1];
            sim_lt[i] = e->refs[i].lt_idx; }
        int any_short = 0; for (int i = 0; i < e->nrefs; i++) if (i != cur_store) any_short |= sim[i][0] == 1 || sim[i][1] == 1;
        const int must = !idr && !second && !any_short && n_used >= p->num_ref;
#define ADD_OP(o, a_, b_) do { e->mmco_op[e->n_mmco] = (o); e->mmco_a[e->n_mmco] = (a_); e->mmco_b[e->n_mmco] = (b_); e->n_mmco++; } while (0)
#define DROP_FIELD(i_, q_) do { Frame *f_ = &e->refs[i_].fld[q_]; if (sim[i_][q_] == 1) ADD_OP(1, cur_pn - PN(f_) - 1, 0); \
        else ADD_OP(2, 2 * sim_lt[i_] + ((q_) == par), 0);      /* (the index may be one an earlier operation of this picture assigned) */ \
        sim[i_][q_] = 0; } while (0)
        const int follow_long = second && e->pending_long_idx >= 0;
        if (!second) e->pending_long_idx = -1;
#define FREE_IDX(ix_, keep_) do { for (int j_ = 0; j_ < e->nrefs; j_++) if (j_ != (keep_) && sim_lt[j_] == (ix_)) for (int q_ = 0; q_ < 2; q_++) \
        if (sim[j_][q_] == 2) sim[j_][q_] = 0; } while (0)
        if (follow_long) { ADD_OP(6, 0, e->pending_long_idx); }          /* (the index is the first field's: nothing else holds it) */
        if (!idr && (must || follow_long || rnd_n(&e->rng, 3) == 0)) {
            if (!follow_long && p->mmco && rnd_n(&e->rng, 3) == 0) {
                /* a store with two short-term fields becomes a long-term pair */
                int full[5], nf = 0;
                for (int i = 0; i < e->nrefs; i++) if (i != cur_store && sim[i][0] == 1 && sim[i][1] == 1 && !e->refs[i].nonexist) full[nf++] = i;
                if (nf > 0) {
                    if (maxlt < 1) { ADD_OP(4, 2, 0); maxlt = 1; }
                    const int i = full[rnd_n(&e->rng, nf)], ix = rnd_n(&e->rng, maxlt + 1);
                    FREE_IDX(ix, i);
                    for (int q = 0; q < 2; q++) { ADD_OP(3, cur_pn - PN(&e->refs[i].fld[q]) - 1, ix); sim[i][q] = 2; }
                    sim_lt[i] = ix;
                }
            } else if (!follow_long && !second && p->mmco && rnd_n(&e->rng, 4) == 0) {
                if (maxlt < 1) { ADD_OP(4, 2, 0); maxlt = 1; }
                const int ix = rnd_n(&e->rng, maxlt + 1);
                FREE_IDX(ix, -1);
                ADD_OP(6, 0, ix); e->pending_long_idx = ix;
            }
            int cand[10][2], nc = 0;
            for (int i = 0; i < e->nrefs; i++) if (i != cur_store) for (int q = 0; q < 2; q++) if (sim[i][q] == 1) { cand[nc][0] = i; cand[nc][1] = q; nc++; }
            if (nc > 0 && rnd_n(&e->rng, 2)) { int k = rnd_n(&e->rng, nc); DROP_FIELD(cand[k][0], cand[k][1]); }
            else if (nc > 0 && rnd_n(&e->rng, 2)) { int i = cand[rnd_n(&e->rng, nc)][0]; for (int q = 0; q < 2; q++) if (sim[i][q]) DROP_FIELD(i, q); }
            if (!second) for (;;) {                                      /* room for the store this field opens */
                int used = 0, old = -1;
                for (int i = 0; i < e->nrefs; i++) if (sim[i][0] || sim[i][1]) { used++;
                    if (old < 0 || ((sim[old][0] == 2 || sim[old][1] == 2) && sim[i][0] != 2 && sim[i][1] != 2) ||
                        (((sim[old][0] == 2 || sim[old][1] == 2) == (sim[i][0] == 2 || sim[i][1] == 2)) && PICNUM(&e->refs[i]) < PICNUM(&e->refs[old]))) old = i; }
                if (used + 1 <= p->num_ref) break;
                for (int q = 0; q < 2; q++) if (sim[old][q]) DROP_FIELD(old, q);
            }
            if (e->n_mmco == 0 && must) { fprintf(stderr, "h264gen: no way to make room\n"); abort(); }
        }
    } else if (p->mmco && is_ref && !is_b && !(half_stores && !idr)) {
        if (idr) e->idr_long = rnd_n(&e->rng, 3) == 0;
        else {
            int st_fn[5], st_n = 0, lt_ix[5], lt_n = 0, maxlt = e->max_lt_idx, cur_long = 0;      /* simulated state */
            for (int i = 0; i < e->nrefs; i++) { if (e->refs[i].is_long) lt_ix[lt_n++] = e->refs[i].lt_idx; else st_fn[st_n++] = PICNUM(&e->refs[i]); }
            /* the sliding window needs a short-term picture to drop (8.2.5.3): with only long-term pictures in a full buffer the
               stream MUST use memory management operations */
            int must = st_n == 0 && lt_n >= p->num_ref;
            /* operation 5 alone.  Not on a picture with frame_num 1: the next picture has frame_num 1 again (7.4.3) and could then differ from this
               one in none of the ways 7.4.1.2.4 lists (same frame_num, both reference pictures, same order-count syntax) */
            if (p->mmco == 2 && curfn != 1 && rnd_n(&e->rng, 5) == 0) { e->mmco_op[0] = 5; e->mmco_a[0] = e->mmco_b[0] = 0; e->n_mmco = 1; }
            else if (must || rnd_n(&e->rng, 3) == 0) {
#define DROP_LT(ix) do { for (int q_ = 0; q_ < lt_n; q_++) if (lt_ix[q_] == (ix)) { lt_ix[q_] = lt_ix[--lt_n]; break; } } while (0)
            if (maxlt < 1 && rnd_n(&e->rng, 2)) { ADD_OP(4, 2, 0); maxlt = 1; }
            if (maxlt >= 0 && st_n > 0 && rnd_n(&e->rng, 2)) { int k = rnd_n(&e->rng, st_n), ix = rnd_n(&e->rng, maxlt + 1), ne = 0;
                for (int i = 0; i < e->nrefs; i++) if (!e->refs[i].is_long && e->refs[i].nonexist && PICNUM(&e->refs[i]) == st_fn[k]) ne = 1;
                if (!ne) { ADD_OP(3, curfn - st_fn[k] - 1, ix); DROP_LT(ix); lt_ix[lt_n++] = ix; st_fn[k] = st_fn[--st_n]; } }   /* (never a frame that was not sent) */
            if (st_n > 0 && rnd_n(&e->rng, 3) == 0) { int k = rnd_n(&e->rng, st_n); ADD_OP(1, curfn - st_fn[k] - 1, 0); st_fn[k] = st_fn[--st_n]; }
            if (lt_n > 0 && rnd_n(&e->rng, 4) == 0) { int k = rnd_n(&e->rng, lt_n); ADD_OP(2, lt_ix[k], 0); lt_ix[k] = lt_ix[--lt_n]; }
            if (maxlt >= 0 && rnd_n(&e->rng, 4) == 0) { int ix = rnd_n(&e->rng, maxlt + 1); ADD_OP(6, 0, ix); DROP_LT(ix); cur_long = 1; }
            (void)cur_long;
            while (st_n + lt_n + 1 > p->num_ref) {                       /* room for the current picture */
                if (st_n > 0) { int k = 0; for (int i = 1; i < st_n; i++) if (st_fn[i] < st_fn[k]) k = i; ADD_OP(1, curfn - st_fn[k] - 1, 0);
                    st_fn[k] = st_fn[--st_n]; }
                else { ADD_OP(2, lt_ix[0], 0); lt_ix[0] = lt_ix[--lt_n]; }
            }
            }
        }
    }
    /* explicit weights of this picture (8.4.2.3); the same table is sent in every slice */
    e->wlog[0] = 5; e->wlog[1] = 5;
    const int nwp = e->field ? 16 : 5;           /* entries of the weight tables drawn (a field list holds up to twice as many entries) */
    for (int l = 0; l < 2; l++) for (int i = 0; i < 16; i++) for (int c = 0; c < 3; c++) { e->ww[l][i][c] = 32; e->wo[l][i][c] = 0; }
    int use_wp = (e->slice_type == 0 && p->wp == 1) || (e->slice_type == 1 && p->wp == 1);
    if (use_wp) {
        e->wlog[0] = 3 + rnd_n(&e->rng, 4); e->wlog[1] = 2 + rnd_n(&e->rng, 4);
        for (int l = 0; l < 2; l++) for (int i = 0; i < nwp; i++) for (int c = 0; c < 3; c++) {
            int one = 1 << e->wlog[c != 0];
            e->ww[l][i][c] = one; e->wo[l][i][c] = 0;
            if (rnd_n(&e->rng, 3)) { e->ww[l][i][c] = one + rnd_n(&e->rng, one / 2 + 1) - one / 4; e->wo[l][i][c] = rnd_n(&e->rng, 13) - 6; }
        }
        for (int l = 0; l < 2; l++) for (int i = 0; i < nwp; i++) if (e->ww[l][i][1] == (1 << e->wlog[1]) && e->wo[l][i][1] == 0 &&
            (e->ww[l][i][2] != (1 << e->wlog[1]) || e->wo[l][i][2] != 0)) e->wo[l][i][1] = 1;
        /* one chroma flag covers Cb and Cr */
    }
    if (!field) { e->cur.id = e->next_id++; e->cur.nonexist = 0; }
    else { if (!second) { full_cur.id = e->next_id++; full_cur.nonexist = 0; } e->cur.id = (1 << 20) + 2 * full_cur.id + par; e->cur.nonexist = 0; }
    int mbs_total = e->mbw * e->mbh, rows_per = (e->mbh + p->slices - 1) / p->slices;
    for (int i = 0; i < mbs_total; i++) e->mbs[i].slice = -1;
    for (int sl = 0, first_row = 0; first_row < e->mbh; sl++, first_row += rows_per) {
        int last_row = MIN(e->mbh, first_row + rows_per);
        e->slice_id = sl; e->qp_run = p->qp;
        w->len = 0; w->nbits = 0; w->cur = 0;
        bw_ue(w, first_row * e->mbw);
        bw_ue(w, e->slice_type + ((sl & 1) ? 0 : 5));                     /* alternate slice_type / slice_type+5 spelling */
        bw_ue(w, 0);
        bw_put(w, e->log2_max_fn, e->frame_num & ((1 << e->log2_max_fn) - 1));
        if (p->fmo0) { bw_put(w, 1, (uint32_t)(field != 0)); if (field) bw_put(w, 1, (uint32_t)(field == 2)); }    /* field_pic_flag, bottom_field_flag */
        if (idr) bw_ue(w, e->idr_id & 0xffff);
        /* a field carries its own count; delta_pic_order_cnt_bottom / delta_pic_order_cnt[1] belong to frames (7.3.3) */
        if (p->poc_type == 0) { bw_put(w, e->poc_lsb_bits, (uint32_t)(field ? field_poc : e->cur_top) & ((1u << e->poc_lsb_bits) - 1));
            if (p->poc_bottom && !field) bw_se(w, e->delta_bottom); }
        if (p->poc_type == 1) { bw_se(w, e->delta0); if (p->poc_bottom && !field) bw_se(w, e->delta_bottom); }
        int rpc_bit = -1;
        if (p->redundant) { rpc_bit = bw_bitpos(w); bw_ue(w, 0); }                              /* redundant_pic_cnt */
        if (e->slice_type == 1) bw_put(w, 1, (uint32_t)!p->direct_temporal);                  /* direct_spatial_mv_pred_flag */
        if (e->slice_type == 0) { int ovr = e->nlist0 != p->num_ref; bw_put(w, 1, ovr); if (ovr) bw_ue(w, e->nlist0 - 1); }
        if (e->slice_type == 1) { bw_put(w, 1, 1); bw_ue(w, e->nlist0 - 1); bw_ue(w, e->nlist1 - 1); }
        for (int l = 0; l < (e->slice_type == 2 ? 0 : (e->slice_type == 1 ? 2 : 1)); l++) {                       /* ref_pic_list_modification() */
            bw_put(w, 1, (uint32_t)(e->n_mod[l] > 0));
            if (e->n_mod[l]) { for (int k = 0; k < e->n_mod[l]; k++) { bw_ue(w, e->mod_idc[l][k]); bw_ue(w, e->mod_val[l][k]); } bw_ue(w, 3); }
        }
        if (use_wp) {                                                     /* pred_weight_table() */
            bw_ue(w, e->wlog[0]); bw_ue(w, e->wlog[1]);
            for (int l = 0; l < (e->slice_type == 1 ? 2 : 1); l++) for (int i = 0; i < (l ? e->nlist1 : e->nlist0); i++) {
                int fy = e->ww[l][i][0] != (1 << e->wlog[0]) || e->wo[l][i][0] != 0;
                int fc = e->ww[l][i][1] != (1 << e->wlog[1]) || e->wo[l][i][1] != 0 || e->ww[l][i][2] != (1 << e->wlog[1]) || e->wo[l][i][2] != 0;
                bw_put(w, 1, (uint32_t)fy); if (fy) { bw_se(w, e->ww[l][i][0]); bw_se(w, e->wo[l][i][0]); }
                bw_put(w, 1, (uint32_t)fc); if (fc) for (int c = 1; c < 3; c++) { bw_se(w, e->ww[l][i][c]); bw_se(w, e->wo[l][i][c]); }
            }
        }
        if (is_ref) {                                                     /* dec_ref_pic_marking() */
            if (idr) { bw_put(w, 1, 0); bw_put(w, 1, (uint32_t)e->idr_long); }
            else { bw_put(w, 1, (uint32_t)(e->n_mmco > 0));
                for (int k = 0; k < e->n_mmco; k++) { int o = e->mmco_op[k]; bw_ue(w, o); if (o == 1 || o == 3) bw_ue(w, e->mmco_a[k]);
                    if (o == 2) bw_ue(w, e->mmco_a[k]); if (o == 3 || o == 6) bw_ue(w, e->mmco_b[k]); if (o == 4) bw_ue(w, e->mmco_a[k]); }
                if (e->n_mmco) bw_ue(w, 0); }
        }
        if (e->cabac && e->slice_type != 2) bw_ue(w, p->cabac_idc);
        bw_se(w, 0);                                                      /* slice_qp_delta */
        int idc = p->deblock == 1 ? 0 : (p->deblock == 0 ? 1 : 2);
        bw_ue(w, idc); if (idc != 1) { bw_se(w, p->alpha_off); bw_se(w, p->beta_off); }
        if (p->redundant && sl == 0) {
            /* the header of the picture's first slice, bit by bit, for the slice of a redundant coded picture that may follow the picture (below) */
            e->hdr_bits = MIN(bw_bitpos(w), (int)sizeof e->hdr_copy * 8); e->hdr_rpc = rpc_bit;
            memset(e->hdr_copy, 0, sizeof e->hdr_copy);
            for (int i = 0; i < e->hdr_bits; i++) if (bw_bit_at(w, i)) e->hdr_copy[i >> 3] |= (uint8_t)(0x80 >> (i & 7));
        }
        int skip_run = e->slice_type != 2 ? 0 : -1;
        if (e->cabac) {
            while (w->nbits) bw_put(w, 1, 1);                             /* cabac_alignment_one_bit */
            cab_init_ctx(&e->cab, e->slice_type == 2 ? 0 : 1 + p->cabac_idc, p->qp);
            cab_start(&e->cab, w); e->last_dqp = 0;
        }
        for (int my = first_row; my < last_row; my++) for (int mx = 0; mx < e->mbw; mx++) {
            MbE *m = &e->mbs[my * e->mbw + mx];
            mb_init(e, m);
            int before = bw_bitpos(w);
            if (p->pcm_only) { se_begin_mb(e, mx, my, &skip_run); encode_intra_mb(e, mx, my, m, 7); }
            else if (e->slice_type == 2) { int force = -1; if (p->mode == 1 && rnd_n(&e->rng, 40) == 0) force = 7; encode_intra_mb(e, mx, my, m, force); }
            else if (e->slice_type == 1) encode_b_mb(e, mx, my, m, &skip_run);
            else encode_p_mb(e, mx, my, m, &skip_run);
            if (e->cabac) cab_term(&e->cab, my == last_row - 1 && mx == e->mbw - 1);      /* end_of_slice_flag */
            e->stat_bits_mb[m->intra ? (m->pcm ? 3 : (m->i16 ? 2 : 1)) : (m->skip ? 4 : 0)] += bw_bitpos(w) - before;
        }
        if (e->cabac) { while (w->nbits) bw_put(w, 1, 0); }              /* the flush ended with the rbsp_stop_one_bit */
        else { if (skip_run > 0) bw_ue(w, skip_run); bw_trailing(w); }
        out_nal(&e->out, is_ref ? (idr ? 3 : 2) : 0, idr ? 5 : 1, w, sl == 0);
    }
    if (p->redundant && e->hdr_rpc >= 0 && rnd_n(&e->rng, 2) == 0) {
        /* a slice of a redundant coded picture (7.4.1.2.4: behind the primary picture's slices, same access unit): the first slice's header with
           redundant_pic_cnt = 1 or 2, and a payload that is NOT slice data -- whoever tries to decode it ends up with a different picture */
        w->len = 0; w->nbits = 0; w->cur = 0;
        for (int i = 0; i < e->hdr_rpc; i++) bw_put(w, 1, (uint32_t)((e->hdr_copy[i >> 3] >> (7 - (i & 7))) & 1));
        bw_ue(w, 1 + (uint32_t)rnd_n(&e->rng, 2));
        for (int i = e->hdr_rpc + 1; i < e->hdr_bits; i++) bw_put(w, 1, (uint32_t)((e->hdr_copy[i >> 3] >> (7 - (i & 7))) & 1));
        for (int i = 0, n = 4 + rnd_n(&e->rng, 24); i < n; i++) bw_put(w, 8, 1 + (uint32_t)rnd_n(&e->rng, 254));
        bw_trailing(w);
        out_nal(&e->out, is_ref ? (idr ? 3 : 2) : 0, idr ? 5 : 1, w, 0);
    }
    if (p->deblock != 0) deblock_frame(e);
#define REMOVE_REF(i_) do { Frame t_ = e->refs[i_]; \
        for (int q_ = (i_); q_ + 1 < e->nrefs; q_++) e->refs[q_] = e->refs[q_ + 1]; \
        e->nrefs--; e->refs[e->nrefs] = t_; } while (0)
    if (field) {
        /* ---- end of a field picture ---- */
        if (getenv("H264GEN_STATS")) {
            fprintf(stderr, "frame %d %s field (%s) type %c ref %d fn %d list0:", t, field == 1 ? "top" : "bottom", second ? "second" : "first",
                    idr ? 'I' : (e->slice_type == 2 ? 'i' : 'P'), is_ref, curfn);
            for (int i = 0; i < e->nlist0; i++) fprintf(stderr, " %s%d%c", e->list0[i]->is_long ? "L" : "fn", e->list0[i]->is_long ? e->list0[i]->lt_idx :
                e->list0[i]->frame_num, e->list0[i]->parity ? 'b' : 't');
            for (int k = 0; k < e->n_mmco; k++) fprintf(stderr, " mmco%d(%d)", e->mmco_op[k], e->mmco_a[k]);
            fprintf(stderr, " bytes %zu\n", e->out.len);
        }
        if (is_ref) {
            frame_finish_ref(&e->cur, e->W, e->H);
            if (p->bframes) {                               /* the field's motion, colocated data of later B fields */
                if (!e->cur.mf) e->cur.mf = malloc(sizeof(MbE) * (size_t)mbs_total);
                memcpy(e->cur.mf, e->mbs, sizeof(MbE) * (size_t)mbs_total);
            }
        }
        full_cur.fld[par] = e->cur; e->src = full_src; e->H *= 2; e->mbh *= 2;
        full_cur.fpoc[par] = field_poc;
        if (is_ref) {
            /* 8.2.5 for a reference field.  The store of the first field is not in e->refs yet; that of the second is e->refs[cur_store] */
            if (cur_store >= 0) e->refs[cur_store] = full_cur;
            if (e->n_mmco) {
                for (int k = 0; k < e->n_mmco; k++) {
                    int o = e->mmco_op[k], a = e->mmco_a[k], b = e->mmco_b[k], hit = 0;
                    if (o == 4) { e->max_lt_idx = a - 1; for (int i = 0; i < e->nrefs; i++) for (int q = 0; q < 2; q++) if (e->refs[i].fmark[q] == 2 &&
                        e->refs[i].lt_idx > e->max_lt_idx) e->refs[i].fmark[q] = 0; continue; }
                    if (o == 3 || o == 6) {
                        /* the index is taken away from every long-term field that is not the partner of the field that gets it (8.2.5.4.3 / 8.2.5.4.6) */
                        int owner = cur_store;                                           /* operation 6: the store of the current frame (-1: not open yet) */
                        if (o == 3) { owner = -1; for (int i = 0; i < e->nrefs; i++) for (int q = 0; q < 2; q++) if (e->refs[i].fmark[q] == 1 &&
                            PN(&e->refs[i].fld[q]) == cur_pn - (a + 1)) { owner = i; e->refs[i].fmark[q] = 2; hit = 1; }
                            if (!hit) { fprintf(stderr, "h264gen: operation 3 names a missing field\n"); abort(); } }
                        for (int i = 0; i < e->nrefs; i++) if (i != owner && e->refs[i].lt_idx == b) for (int q = 0; q < 2; q++) if (e->refs[i].fmark[q] == 2)
                            e->refs[i].fmark[q] = 0;
                        if (o == 3) e->refs[owner].lt_idx = b;
                        continue;
                    }
                    for (int i = 0; i < e->nrefs && !hit; i++) for (int q = 0; q < 2 && !hit; q++) {
                        Frame *f = &e->refs[i].fld[q];
                        if ((o == 1 && e->refs[i].fmark[q] == 1 && PN(f) == cur_pn - (a + 1)) || (o == 2 && e->refs[i].fmark[q] == 2 && 2 * e->refs[i].lt_idx + (q == par) == a)) {
                            e->refs[i].fmark[q] = 0; hit = 1;
                            if (!e->refs[i].fmark[0] && !e->refs[i].fmark[1]) { if (i == cur_store) abort(); REMOVE_REF(i); if (cur_store > i) cur_store--; }
                        }
                    }
                    if (!hit) { fprintf(stderr, "h264gen: field MMCO names a missing picture\n"); abort(); }
                }
                for (int i = e->nrefs - 1; i >= 0; i--) {                    /* stores left without a reference field go; whole long-term pairs are frames */
                    if (!e->refs[i].fmark[0] && !e->refs[i].fmark[1]) { if (i == cur_store) abort(); REMOVE_REF(i); if (cur_store > i) cur_store--; continue; }
                    e->refs[i].is_long = e->refs[i].fmark[0] == 2 && e->refs[i].fmark[1] == 2;
                }
            } else if (!second && !idr && e->nrefs >= p->num_ref) {       /* sliding window (8.2.5.3); never for the second field of a reference frame */
                int old = -1;
                for (int i = 0; i < e->nrefs; i++) if ((e->refs[i].fmark[0] == 1 || e->refs[i].fmark[1] == 1) && (old < 0 ||
                    PICNUM(&e->refs[i]) < PICNUM(&e->refs[old]))) old = i;
                if (old < 0) { fprintf(stderr, "h264gen: sliding window without a short-term picture\n"); abort(); }
                REMOVE_REF(old);
            }
            if (!second) {
                if (idr) e->max_lt_idx = -1;
                if (e->nrefs >= p->num_ref) { fprintf(stderr, "h264gen: no free frame store for a field\n"); abort(); }
                full_cur.frame_num = curfn; full_cur.is_long = 0; full_cur.lt_idx = e->pending_long_idx;
                full_cur.fmark[par] = e->pending_long_idx >= 0 ? 2 : 1; full_cur.fmark[par ^ 1] = 0;
                full_cur.poc = field_poc; full_cur.coded_fields = 1;
                e->cur_store_id = full_cur.id;
                Frame t_ = e->refs[e->nrefs]; e->refs[e->nrefs] = full_cur; e->cur = t_; e->nrefs++;
            } else {
                Frame *s = &e->refs[cur_store];
                s->fmark[par] = e->pending_long_idx >= 0 ? 2 : 1; s->is_long = s->fmark[0] == 2 && s->fmark[1] == 2; s->poc = MIN(s->fpoc[0], s->fpoc[1]);
                e->pending_long_idx = -1;
                for (int q = 0; q < 2; q++) field_copy(s, &s->fld[q], q, e->W, e->H, 0);
                frame_finish_ref(s, e->W, e->H);
                e->frame_num++;
                e->cur = spare_cur;
            }
        } else {
            e->cur = full_cur;
            if (second) for (int q = 0; q < 2; q++) field_copy(&e->cur, &e->cur.fld[q], q, e->W, e->H, 0);
        }
        if (second && e->recon_buf && t < e->recon_frames) {
            const Frame *s = is_ref ? &e->refs[cur_store] : &e->cur;
            uint8_t *o = e->recon_buf + (size_t)t * (p->width * p->height * 3 / 2);
            for (int y = 0; y < p->height; y++) memcpy(o + (size_t)y * p->width, s->y + y * s->sy, p->width);
            o += (size_t)p->width * p->height;
            for (int y = 0; y < p->height / 2; y++) memcpy(o + (size_t)y * (p->width / 2), s->u + y * s->sc, p->width / 2);
            o += (size_t)(p->width / 2) * (p->height / 2);
            for (int y = 0; y < p->height / 2; y++) memcpy(o + (size_t)y * (p->width / 2), s->v + y * s->sc, p->width / 2);
        }
        if (idr) e->idr_id++;
        return;
    }
    if (getenv("H264GEN_STATS")) {
        double se = 0; int cnt[6] = {0}, nzmv = 0, qmv = 0;
        for (int y = 0; y < p->height; y++) for (int x = 0; x < p->width; x++) { int d = e->src.y[y * e->src.sy + x] - e->cur.y[y * e->cur.sy + x];
            se += d * d; }
        for (int i = 0; i < mbs_total; i++) { MbE *m = &e->mbs[i]; cnt[m->intra ? (m->pcm ? 3 : (m->i16 ? 2 : 1)) : (m->skip ? 4 : 0)]++;
            if (!m->intra) for (int k = 0; k < 16; k++) { nzmv += m->mv[k][0] || m->mv[k][1]; qmv += (m->mv[k][0] & 3) || (m->mv[k][1] & 3); } }
        double mse = se / (p->width * p->height);
        if (e->n_mmco || e->n_mod[0] || e->n_mod[1]) { fprintf(stderr, "  frame %d fn %d:", t, e->frame_num);
            for (int k = 0; k < e->n_mmco; k++) fprintf(stderr, " mmco%d(%d,%d)", e->mmco_op[k], e->mmco_a[k], e->mmco_b[k]);
            for (int l = 0; l < 2; l++) for (int k = 0; k < e->n_mod[l]; k++) fprintf(stderr, " mod%d(%d,%d)", l, e->mod_idc[l][k], e->mod_val[l][k]);
            fprintf(stderr, " list0:"); for (int i = 0; i < e->nlist0; i++) fprintf(stderr, " %s%d", e->list0[i]->is_long ? "L" : "fn",
            e->list0[i]->is_long ? e->list0[i]->lt_idx : e->list0[i]->frame_num); fprintf(stderr, "\n"); }
        fprintf(stderr, "frame %d type %c ref %d mse %.2f inter %d i4 %d i16 %d pcm %d skip %d nzmv4x4 %d qpelmv4x4 %d bytes %zu\n", t,
            idr ? 'I' : (e->slice_type == 2 ? 'i' : (is_b ? 'B' : 'P')), is_ref, mse, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], nzmv, qmv, e->out.len);
    }
    if (e->recon_buf && t < e->recon_frames) {                           /* display order */
        uint8_t *o = e->recon_buf + (size_t)t * (p->width * p->height * 3 / 2);
        for (int y = 0; y < p->height; y++) memcpy(o + (size_t)y * p->width, e->cur.y + y * e->cur.sy, p->width);
        o += (size_t)p->width * p->height;
        for (int y = 0; y < p->height / 2; y++) memcpy(o + (size_t)y * (p->width / 2), e->cur.u + y * e->cur.sc, p->width / 2);
        o += (size_t)(p->width / 2) * (p->height / 2);
        for (int y = 0; y < p->height / 2; y++) memcpy(o + (size_t)y * (p->width / 2), e->cur.v + y * e->cur.sc, p->width / 2);
    }
    if (is_ref) {
        frame_finish_ref(&e->cur, e->W, e->H);
        if (!e->cur.mf) e->cur.mf = malloc(sizeof(MbE) * (size_t)mbs_total);
        memcpy(e->cur.mf, e->mbs, sizeof(MbE) * (size_t)mbs_total);
        /* 8.2.5 marking; e->refs[] is the set of reference frames, refs[nrefs..4] + cur are free frame stores */
        if (idr) { e->max_lt_idx = e->idr_long ? 0 : -1; e->cur.is_long = e->idr_long; e->cur.lt_idx = e->idr_long ? 0 : -1; }
        else if (e->n_mmco) {
            for (int k = 0; k < e->n_mmco; k++) {
                int o = e->mmco_op[k], a = e->mmco_a[k], b = e->mmco_b[k];
                if (o == 1 || o == 3) {
                    int pn = curfn - (a + 1), found = -1;
                    for (int i = 0; i < e->nrefs; i++) if (!e->refs[i].is_long && PICNUM(&e->refs[i]) == pn) found = i;
                    if (found < 0) { fprintf(stderr, "h264gen: MMCO names a missing picture\n"); abort(); }
                    if (o == 1) REMOVE_REF(found);
                    else { for (int i = 0; i < e->nrefs; i++) if (i != found && e->refs[i].is_long && e->refs[i].lt_idx == b) { REMOVE_REF(i);
                        if (i < found) found--; break; }
                           e->refs[found].is_long = 1; e->refs[found].lt_idx = b; }
                } else if (o == 2) { for (int i = 0; i < e->nrefs; i++) if (e->refs[i].is_long && e->refs[i].lt_idx == a) { REMOVE_REF(i); break; } }
                else if (o == 4) { e->max_lt_idx = a - 1; for (int i = e->nrefs - 1; i >= 0; i--) if (e->refs[i].is_long &&
                    e->refs[i].lt_idx > e->max_lt_idx) REMOVE_REF(i); }
                else if (o == 6) { for (int i = 0; i < e->nrefs; i++) if (e->refs[i].is_long && e->refs[i].lt_idx == b) { REMOVE_REF(i); break; }
                    e->cur.is_long = 1; e->cur.lt_idx = b; }
                else if (o == 5) {
                    /* every reference picture is dropped; the picture is inferred to have had frame_num 0 (7.4.3) and its order counts are
                       reduced by Min(top, bottom) (8.2.1): what follows counts from here */
                    while (e->nrefs > 0) REMOVE_REF(0);
                    e->max_lt_idx = -1; e->cur.frame_num = 0; e->cur.poc = 0; e->frame_num = 0; e->poc_base = t; e->after_op5 = 1;
                    if (e->pocs) e->pocs[t] = 0;
                }
            }
        } else if (e->nrefs >= p->num_ref) {                              /* sliding window: drop the oldest short-term picture */
            int old = -1;
            for (int i = 0; i < e->nrefs; i++) if (!e->refs[i].is_long && (old < 0 || PICNUM(&e->refs[i]) < PICNUM(&e->refs[old]))) old = i;
            if (old < 0) old = 0;
            REMOVE_REF(old);
        }
        if (p->paff) {
            /* the frame's two fields as pictures of their own, for the field pictures that follow */
            for (int i = 0; i < e->nrefs; i++) if (e->refs[i].is_long) e->refs[i].fmark[0] = e->refs[i].fmark[1] = 2;
            e->cur.fmark[0] = e->cur.fmark[1] = e->cur.is_long ? 2 : 1; e->cur.coded_fields = 0;
            e->cur.fpoc[0] = e->cur.poc - MIN(0, e->delta_bottom); e->cur.fpoc[1] = e->cur.fpoc[0] + e->delta_bottom;
            for (int q = 0; q < 2; q++) { field_copy(&e->cur, &e->cur.fld[q], q, e->W, e->H, 1); frame_finish_ref(&e->cur.fld[q], e->W, e->H / 2); }
        }
        { Frame t_ = e->refs[e->nrefs]; e->refs[e->nrefs] = e->cur; e->cur = t_; e->nrefs++; }
        e->frame_num++;
    }
    if (idr) e->idr_id++;
}

/* one frame of the stream: a frame picture, or (PAFF) two field pictures -- the first of either parity */
static void encode_frame(Enc *e, int t, int is_b) {
    int as_fields = e->p.paff && (e->p.paff == 2 || rnd_n(&e->rng, 2));
    if (!as_fields) { encode_picture(e, t, is_b, 0, 0); return; }
    int first = 1 + rnd_n(&e->rng, 2);
    encode_picture(e, t, is_b, first, 0);
    encode_picture(e, t, is_b, 3 - first, 1);
}

/* The picture order counts the encoder MEANT, by display index, of the stream this thread generated last: TopFieldOrderCnt counts 2 per picture from
   the last IDR picture or operation 5, PicOrderCnt = Min(top, bottom), 0 for a picture that carried operation 5 (8.2.1).  For pic_order_cnt_type 0 a
   decoder must reconstruct exactly these values from pic_order_cnt_lsb / delta_pic_order_cnt_bottom (tests/test_host_parser.py); types 1 and 2
   derive other (equally ordered) values.  Returns the number of pictures. */
static __thread int *g_last_pocs; static __thread int g_last_n;
int h264gen_last_pocs(int *buf, int max) { for (int i = 0; i < g_last_n && i < max; i++) buf[i] = g_last_pocs[i]; return g_last_n; }
/* library entry: returns malloc'ed Annex-B stream */
int h264gen_generate(const GenParams *gp, uint8_t **out, size_t *out_len, const char *recon_path) {
    g_nc_corner = gp->nc_corner;
    Enc *e = (Enc *)calloc(1, sizeof(Enc));
    e->p = *gp;
    GenParams *p = &e->p;
    if (p->width < 16 || p->height < 16 || (p->width & 1) || (p->height & 1)) { free(e); return -1; }
    if (p->gop < 1) p->gop = 30;
    if (p->num_ref < 1) p->num_ref = 1;
    if (p->num_ref > 4) p->num_ref = 4;
    if (p->slices < 1) p->slices = 1;
    if (p->search < 1) p->search = 4;
    if (!p->level_idc) p->level_idc = 40;
    if (p->poc_type != 0 && p->poc_type != 1) p->poc_type = 2;
    p->poc_bottom = p->poc_bottom != 0;
    p->scaling = CLIP3(0, 2, p->scaling);
    p->paff = CLIP3(0, 2, p->paff);
    if (p->paff) { p->fmo0 = 1; if (p->cabac) p->t8x8 = 0; if (p->wp == 2 && !p->bframes) p->wp = 0; }
    if (p->bframes) { p->mmco = 0; p->gaps = 0; }
    p->gaps = p->gaps != 0; p->redundant = p->redundant != 0;
    e->max_lt_idx = -1; e->pending_long_idx = -1;
    p->bframes = CLIP3(0, 3, p->bframes); p->wp = CLIP3(0, 2, p->wp); p->direct_temporal = p->direct_temporal != 0;
    if (!p->bframes) { if (p->wp == 2) p->wp = 0; p->dinf8 = 1; }
    else { p->poc_type = 0; p->nonref_period = 0; if (p->num_ref < 2) p->num_ref = 2; p->dinf8 = p->dinf8 != 0; }
    /* pic_order_cnt_type 1 (8.2.1.2): reference pictures advance by 8 / 10 / 12 in a cycle of 1..3, a non-reference picture lies 4 above the
       reference picture before it, top-to-bottom offset -1..1; with delta_pic_order_cnt[0] in 0..1 and the bottom deltas in -1..1 the counts still
       rise strictly in decoding order, which is the display order of a stream without B pictures */
    { uint32_t h = (uint32_t)p->seed * 2654435761u; e->t1_cycle = 1 + (int)((h >> 8) % 3);
        for (int i = 0; i < 3; i++) e->t1_ref[i] = 8 + 2 * (int)((h >> (12 + 4 * i)) % 3);
      e->t1_nonref = 4; e->t1_t2b = p->poc_bottom ? (int)((h >> 26) % 3) - 1 : 0; }
    p->cabac = p->cabac != 0; p->t8x8 = p->t8x8 != 0; p->cabac_idc = CLIP3(0, 2, p->cabac_idc);
    e->cabac = p->cabac;
    if (p->nonref_period == 1) p->nonref_period = 2;
    e->mbw = (p->width + 15) / 16; e->mbh = (p->height + 15) / 16; e->W = e->mbw * 16; e->H = e->mbh * 16;
    p->fmo0 = p->fmo0 != 0;
    if (p->fmo0) { if ((e->mbh & 1) || ((e->H - p->height) & 3)) { free(e); return -1; } p->dinf8 = 1; }
    if (p->slices > e->mbh) p->slices = e->mbh;
    e->log2_max_fn = 4 + (p->seed & 3); e->poc_lsb_bits = 6 + (p->seed & 1) * 2;
    e->rng.s = (uint64_t)p->seed * 0x9E3779B97F4A7C15ull + 12345;
    frame_alloc(&e->src, e->W, e->H, 0); frame_alloc(&e->cur, e->W, e->H, 1);
    for (int i = 0; i < 5; i++) frame_alloc(&e->refs[i], e->W, e->H, 1);
    if (p->paff) {
        frame_alloc(&e->fsrc, e->W, e->H / 2, 0);
        for (int i = 0; i < 6; i++) { Frame *f = i < 5 ? &e->refs[i] : &e->cur; f->fld = (Frame *)calloc(2, sizeof(Frame));
            for (int q = 0; q < 2; q++) frame_alloc(&f->fld[q], e->W, e->H / 2, 1); }
    }
    e->mbs = (MbE *)calloc((size_t)e->mbw * e->mbh, sizeof(MbE));
    make_texture(e);
    e->pocs = (int *)calloc((size_t)p->frames + 1, sizeof(int));
    if (recon_path) { e->recon = fopen(recon_path, "wb"); e->recon_frames = p->frames;
        e->recon_buf = (uint8_t *)calloc((size_t)p->frames, (size_t)p->width * p->height * 3 / 2); }
    for (int g0 = 0; g0 < p->frames; g0 += p->gop) {
        /* coding order: every anchor before the B pictures that precede it in display order */
        int g1 = MIN(p->frames, g0 + p->gop), prev = g0;
        encode_frame(e, g0, 0);
        while (prev + 1 < g1) {
            int anchor = MIN(g1 - 1, prev + p->bframes + 1);
            encode_frame(e, anchor, 0);
            for (int t = prev + 1; t < anchor; t++) encode_frame(e, t, 1);
            prev = anchor;
        }
    }
    if (e->recon) { fwrite(e->recon_buf, 1, (size_t)p->frames * ((size_t)p->width * p->height * 3 / 2), e->recon); fclose(e->recon); free(e->recon_buf); }
    *out = e->out.buf; *out_len = e->out.len;
    free(g_last_pocs); g_last_pocs = e->pocs; g_last_n = p->frames;
    free(e->cur.mf); for (int i = 0; i < 5; i++) free(e->refs[i].mf);
    if (p->paff) { frame_free(&e->fsrc);
        for (int i = 0; i < 6; i++) { Frame *f = i < 5 ? &e->refs[i] : &e->cur; for (int q = 0; q < 2; q++) { free(f->fld[q].mf); frame_free(&f->fld[q]); }
            free(f->fld); } }
    frame_free(&e->src); frame_free(&e->cur); for (int i = 0; i < 5; i++) frame_free(&e->refs[i]);
    free(e->mbs); free(e->bw.buf); free(e);
    return 0;
}
void h264gen_free(void *p) { free(p); }

#ifndef H264GEN_NO_MAIN
int main(int argc, char **argv) {
    GenParams p; memset(&p, 0, sizeof p);
    p.width = 1920; p.height = 1080; p.frames = 30; p.qp = 28; p.gop = 30; p.seed = 0x4A4D0100; p.deblock = 1; p.num_ref = 1; p.slices = 1; p.search = 4;
    p.dinf8 = 1;
    const char *outp = NULL, *recon = NULL;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i]; const char *v = i + 1 < argc ? argv[i + 1] : "0";
#define OPT(name, field) if (!strcmp(a, name)) { p.field = (int)strtol(v, NULL, 0); i++; continue; }
        OPT("--width", width) OPT("--height", height) OPT("--frames", frames) OPT("--qp", qp) OPT("--gop", gop) OPT("--seed", seed)
        OPT("--mode", mode) OPT("--deblock", deblock) OPT("--refs", num_ref) OPT("--slices", slices) OPT("--pcm", pcm_only)
        OPT("--poc-type", poc_type) OPT("--nonref", nonref_period) OPT("--alpha", alpha_off) OPT("--beta", beta_off)
        OPT("--cqp", chroma_qp_off) OPT("--level", level_idc) OPT("--cip", cip) OPT("--search", search)
        OPT("--cabac", cabac) OPT("--cabac-idc", cabac_idc) OPT("--t8x8", t8x8)
        OPT("--bframes", bframes) OPT("--direct-temporal", direct_temporal) OPT("--wp", wp) OPT("--dinf8", dinf8) OPT("--scaling", scaling) OPT("--rplm",
            rplm) OPT("--mmco", mmco) OPT("--nc-corner", nc_corner) OPT("--no-intra", no_intra) OPT("--fmo0", fmo0) OPT("--poc-bottom", poc_bottom) OPT("--paff", paff) OPT("--gaps", gaps) OPT("--redundant", redundant) OPT("--vui-fps", vui_fps)
        if (!strcmp(a, "-o")) { outp = v; i++; continue; }
        if (!strcmp(a, "--recon")) { recon = v; i++; continue; }
        fprintf(stderr, "unknown option %s\n", a); return 2;
    }
    if (!outp) { fprintf(stderr,
        "usage: h264gen [--width W --height H --frames N --qp Q --gop G --seed S --mode 0|1 --deblock 0|1|2 --refs N --slices N --pcm 1 ...] "
        "-o out.h264 [--recon recon.yuv]\n"); return 2; }
    uint8_t *buf; size_t len;
    if (h264gen_generate(&p, &buf, &len, recon) < 0) { fprintf(stderr, "bad parameters\n"); return 1; }
    FILE *f = fopen(outp, "wb"); fwrite(buf, 1, len, f); fclose(f);
    fprintf(stderr, "wrote %zu bytes, %d frames (%.1f kbit/frame)\n", len, p.frames, len * 8.0 / 1000 / p.frames);
    free(buf);
    return 0;
}
#endif
