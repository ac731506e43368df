/* oracle/orc_internal.h -- CPU ORACLE internals (test infrastructure only). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "orc_h264.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ---------------- bit reader over an RBSP (emulation bytes removed) ------ */
typedef struct {
    const uint8_t *p;
    size_t nbits;   /* total bits in the RBSP payload                      */
    size_t pos;     /* next bit to read                                    */
    int err;        /* set when reading past the end                       */
} Bits;

static inline void bits_init(Bits *b, const uint8_t *p, size_t nbytes) {
    b->p = p; b->nbits = nbytes * 8; b->pos = 0; b->err = 0;
}
static inline unsigned bits_u1(Bits *b) {
    if (b->pos >= b->nbits) { b->err = 1; return 0; }
    unsigned v = (b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1;
    b->pos++;
    return v;
}
static inline unsigned bits_u(Bits *b, int n) {
    unsigned v = 0;
    for (int i = 0; i < n; i++) v = (v << 1) | bits_u1(b);
    return v;
}
static inline unsigned bits_peek(Bits *b, int n) {   /* zero-extends past end */
    unsigned v = 0;
    for (int i = 0; i < n; i++) {
        size_t q = b->pos + i;
        unsigned bit = q < b->nbits ? (b->p[q >> 3] >> (7 - (q & 7))) & 1 : 0;
        v = (v << 1) | bit;
    }
    return v;
}
static inline unsigned bits_ue(Bits *b) {            /* 9.1 Exp-Golomb ue(v)  */
    int lz = 0;
    while (!bits_u1(b)) { if (b->err || ++lz > 32) { b->err = 1; return 0; } }
    if (lz == 0) return 0;
    if (lz == 32) return 0xFFFFFFFFu;
    return (1u << lz) - 1 + bits_u(b, lz);
}
static inline int bits_se(Bits *b) {                 /* 9.1.1 se(v)           */
    unsigned k = bits_ue(b);
    return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1);
}
static inline int bits_te(Bits *b, int range_max) {  /* 9.1 te(v)             */
    if (range_max > 1) return (int)bits_ue(b);
    return !bits_u1(b);
}
static inline int bits_aligned(const Bits *b) { return (b->pos & 7) == 0; }
/* 7.2 more_rbsp_data(): true if there is more before the rbsp_trailing_bits */
static inline int bits_more_rbsp(const Bits *b) {
    if (b->pos >= b->nbits) return 0;
    /* find the last 1 bit in the payload (the stop bit) */
    size_t last = b->nbits;
    while (last > 0) {
        size_t q = last - 1;
        if ((b->p[q >> 3] >> (7 - (q & 7))) & 1) break;
        last--;
    }
    if (last == 0) return 0;
    return b->pos < last - 1;
}

/* ---------------- parameter sets ------------------------------------------ */
typedef struct {
    int valid;
    int profile_idc, constraint_flags, level_idc, sps_id;
    int chroma_format_idc, bit_depth_luma, bit_depth_chroma;
    int qpprime_y_zero_transform_bypass;
    int scaling_matrix_present;
    uint8_t scaling4[6][16];      /* in zig-zag order as transmitted -> stored raster */
    uint8_t scaling8[2][64];
    int log2_max_frame_num, poc_type, log2_max_poc_lsb;
    int delta_pic_order_always_zero, offset_for_non_ref_pic, offset_for_top_to_bottom;
    int num_ref_frames_in_poc_cycle; int offset_for_ref_frame[256];
    int max_num_ref_frames, gaps_in_frame_num_allowed;
    int mb_width, mb_height, frame_mbs_only, mb_aff, direct_8x8_inference;
    int crop, crop_left, crop_right, crop_top, crop_bottom;
    int vui_present, bitstream_restriction, max_num_reorder_frames, max_dec_frame_buffering;
} Sps;

typedef struct {
    int valid;
    int pps_id, sps_id, entropy_coding_mode, bottom_field_pic_order_present;
    int num_slice_groups;
    int num_ref_idx_l0_default, num_ref_idx_l1_default;
    int weighted_pred, weighted_bipred_idc;
    int pic_init_qp, pic_init_qs, chroma_qp_index_offset, second_chroma_qp_index_offset;
    int deblocking_filter_control_present, constrained_intra_pred, redundant_pic_cnt_present;
    int transform_8x8_mode, scaling_matrix_present;
    uint8_t scaling4[6][16];
    uint8_t scaling8[2][64];
} Pps;

enum { SLICE_P = 0, SLICE_B = 1, SLICE_I = 2, SLICE_SP = 3, SLICE_SI = 4 };

typedef struct { int idc; int val; } RplmOp;
typedef struct { int op, diff_pic_nums_minus1, long_term_pic_num, long_term_frame_idx, max_long_term_frame_idx_plus1; } Mmco;

typedef struct {
    int first_mb, slice_type, pps_id, frame_num, idr_pic_id;
    int field_pic, bottom_field;      /* field_pic_flag, bottom_field_flag: the picture is one field of a frame, decoded as a picture of half the height */
    int poc_lsb, delta_poc_bottom, delta_poc[2];
    int redundant_pic_cnt, direct_spatial_mv_pred;
    int num_ref_idx[2];
    int rplm_flag[2]; int n_rplm[2]; RplmOp rplm[2][66];
    int luma_log2_wd, chroma_log2_wd;
    int luma_weight_flag[2][32], chroma_weight_flag[2][32];
    int luma_weight[2][32], luma_offset[2][32], chroma_weight[2][32][2], chroma_offset[2][32][2];
    int no_output_of_prior_pics, long_term_reference_flag, adaptive_marking; int n_mmco; Mmco mmco[66];
    int cabac_init_idc, slice_qp_delta, qp;
    int disable_deblock, alpha_c0_offset, beta_offset;   /* offsets already *2 */
    int nal_ref_idc, idr;
} SliceHdr;

/* ---------------- pictures -------------------------------------------------- */
typedef struct {
    int8_t  ref_idx[2][4];     /* [list] per 8x8 ; -1 = list not used / intra                    */
    int16_t mv[2][16][2];      /* [list] per 4x4 (raster within MB)                              */
    int     ref_pic_id[2][4];  /* picture identity of the reference per 8x8 (deblock bS, direct) */
    uint16_t mvd[2][16][2];    /* |mvd| per 4x4, CABAC ctxIdxInc of mvd (9.3.3.1.1.7)            */
    uint8_t total_coeff[16 + 4 + 4]; /* luma 16 (raster) + Cb 4 + Cr 4                          */
    uint32_t cbf;              /* CABAC coded_block_flag: bits 0-15 luma raster, 16 I16 DC, 17/18 Cb/Cr DC, 19-22 Cb AC, 23-26 Cr AC */
    uint8_t is_intra, is_pcm, is_i16, is_skip, t8x8;
    uint8_t direct8;           /* bit b8: 8x8 quadrant predicted in direct mode (B_Skip / B_Direct_16x16: 0xF) */
    uint8_t b_direct16;        /* B_Skip or B_Direct_16x16 */
    uint8_t chroma_pred_mode;
    uint8_t qp, qpc[2];
    uint8_t i4mode[16];        /* Intra4x4PredMode per 4x4 raster (Intra8x8PredMode replicated over its four 4x4s; 2 = DC default) */
    int16_t slice_num;         /* slice this MB belongs to; -1 = not decoded  */
    uint8_t disable_deblock; int8_t alpha_off, beta_off;
    uint16_t cbp;
    uint8_t mb_type_p;         /* inter partition type: 0 16x16 1 16x8 2 8x16 3 8x8 */
} MbInfo;

typedef struct Picture {
    uint8_t *y, *u, *v;
    int stride_y, stride_c;
    MbInfo *mbs;
    int poc, frame_num, frame_num_wrap, pic_num, long_term_frame_idx, long_term_pic_num;
    int is_ref;          /* 0 none, 1 short-term, 2 long-term */
    int needed_for_output;
    int in_use;          /* slot allocated */
    int id;              /* unique increasing identity */
    int frame_type, decode_index, is_idr;
    int has_mmco5;
    /* Interlace (PAFF).  A Picture in the DPB is a FRAME STORE (C.4.5): a frame, or one or two field pictures of the same frame.  Each field carries
     * its own marking and order count; is_ref is derived from them (orc_sync_ref): 1 / 2 = BOTH fields short- / long-term, i.e. a reference frame in
     * the sense of 8.2.4.2.1; 3 = some field is a reference, which only field pictures can use but which keeps the store occupied. */
    int fmark[2];        /* per field (top, bottom): 0 not a reference, 1 short-term, 2 long-term */
    int fpoc[2];         /* TopFieldOrderCnt, BottomFieldOrderCnt */
    int have;            /* bit 0 / bit 1: the top / bottom field has been decoded (a frame picture decodes both) */
    int waiting_second;  /* holds a first field whose second field may follow as the next picture */
    int first_was_ref;   /* nal_ref_idc != 0 of that first field (3.30: reference fields pair with reference fields) */
    int non_existing;    /* a frame inferred from a gap in frame_num (8.2.5.2): a short-term reference without samples, never output */
    int coded_fields;    /* the store was filled by field pictures (its MbInfo are two field arrays), not by a frame picture */
    /* FIELD VIEW (OrcDec.fview): the lines of one parity of a store as a picture of half the height and twice the stride -- what a field picture
     * is decoded into and predicts from.  is_ref, poc, pic_num, long_term_pic_num of a view are the FIELD's (8.2.4.1). */
    int is_field, parity; struct Picture *store;
} Picture;

#define ORC_MAX_DPB 17

struct OrcDec {
    orc_frame_cb cb; void *user;
    char err[256];
    Sps sps[32]; Pps pps[256];
    const Sps *asps; const Pps *apps;   /* active */
    int mb_w, mb_h, width, height;
    Picture dpb[ORC_MAX_DPB + 1];
    Picture fview[ORC_MAX_DPB + 1][2];
    Picture *cur;         /* what the slices decode into: the frame store itself, or the view of the field being decoded */
    Picture *cur_store;   /* the frame store of the current picture */
    Picture *pending;     /* the store that holds a first field and waits for the second */
    int sticky_fail;      /* a picture ended in something unsupported: every later call fails */
    int field_pic, cur_parity, is_second_field;     /* the current picture is a field picture (mb_h / height are the FIELD's while it is decoded) */
    int dpb_size;
    int next_pic_id, decode_count;
    /* POC state */
    int prev_poc_msb, prev_poc_lsb, prev_frame_num, prev_frame_num_offset, prev_ref_has_mmco5; int cur_top_poc, cur_bot_poc;
    int prev_ref_frame_num;   /* PrevRefFrameNum (7.4.3): frame_num of the previous reference picture (0 after an IDR picture or operation 5) */
    /* TopFieldOrderCnt / BottomFieldOrderCnt of the current picture (pic_order_cnt_type 0) */
    /* slice state */
    SliceHdr sh; SliceHdr first_sh;
    int slice_num;
    Picture *ref_list[2][33]; int ref_count[2];
    int cur_mb_count;     /* MBs decoded in current picture */
    int max_long_term_frame_idx;
    uint8_t *rbsp; size_t rbsp_cap;
    int last_poc_out;
    uint8_t *outbuf; /* crop scratch */
    int digest_on; uint64_t digest; uint64_t digest_mbs;
    long stats[40];       /* ORC_ST_* tool-usage counters (which coding tools a stream exercised) */
};

/* orc_parse.c */
int orc_parse_sps(OrcDec *d, Bits *b);
int orc_parse_pps(OrcDec *d, Bits *b);
int orc_parse_slice_header(OrcDec *d, Bits *b, int nal_unit_type, int nal_ref_idc, SliceHdr *sh);
/* orc_slice.c */
int orc_decode_slice_data(OrcDec *d, Bits *b);
/* orc_deblock.c */
void orc_deblock_picture(OrcDec *d, Picture *pic);
/* orc_dpb.c */
int  orc_start_picture(OrcDec *d, const SliceHdr *sh);
void orc_finish_picture(OrcDec *d);
int  orc_build_ref_lists(OrcDec *d, const SliceHdr *sh);
void orc_sync_ref(Picture *p);
void orc_output_all(OrcDec *d);

enum { ORC_ST_I4, ORC_ST_I8, ORC_ST_I16, ORC_ST_PCM, ORC_ST_PSKIP, ORC_ST_P16, ORC_ST_P16x8, ORC_ST_P8x16, ORC_ST_P8x8, ORC_ST_SUB_SMALL,
       ORC_ST_T8_INTER, ORC_ST_CABAC_SLICES, ORC_ST_CAVLC_SLICES, ORC_ST_IDC0, ORC_ST_IDC1, ORC_ST_IDC2, ORC_ST_MULTIREF, ORC_ST_BSKIP, ORC_ST_BDIRECT,
           ORC_ST_BINTER, ORC_ST_EXACT_END,
       /* interlace: field pictures, second fields, 4x4 blocks predicted from a field of the other parity, memory management operations / list
          modifications / sliding-window removals in field pictures, long-term fields in a field's list, frame pictures that met a store with one
          reference field only, intra macroblock edges that got bS 3 because they run horizontally through a field, vector pairs whose vertical
          difference of 2 or 3 counted only because of the field rule, fields that stayed without partner */
       ORC_ST_FIELD_PICS, ORC_ST_SECOND_FIELDS, ORC_ST_CROSS_PARITY, ORC_ST_FIELD_MMCO, ORC_ST_FIELD_RPLM, ORC_ST_FIELD_WINDOW, ORC_ST_FIELD_LONG,
       ORC_ST_HALF_STORE, ORC_ST_FIELD_BS3, ORC_ST_FIELD_MVY, ORC_ST_LONE_FIELD, ORC_ST_B_FIELDS, ORC_ST_DIRECT_MIXED, ORC_ST_FIELD_LONG_OPS, ORC_ST_INFERRED_FRAMES, ORC_ST_REDUNDANT, ORC_ST_N };

#define ORC_FAIL(d, ...) do { snprintf((d)->err, sizeof((d)->err), __VA_ARGS__); return -1; } while (0)

static inline int orc_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int orc_clip1(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static inline int orc_abs(int v) { return v < 0 ? -v : v; }
static inline int orc_min(int a, int b) { return a < b ? a : b; }
static inline int orc_max(int a, int b) { return a > b ? a : b; }

#endif
