/* oracle/orc_hevc_internal.h -- CPU ORACLE internals for HEVC (test infrastructure only, see orc_hevc.h). */
#ifndef ORC_HEVC_INTERNAL_H
#define ORC_HEVC_INTERNAL_H
#include "orc_hevc.h"
#include "orc_internal.h"      /* Bits reader (shared with the H.264 oracle) */
#include "orc_hevc_tables.h"

#define H_MAX_REFS 16
#define H_MAX_DPB 17
#define H_MAX_SLICES 600

enum { H_SLICE_B = 0, H_SLICE_P = 1, H_SLICE_I = 2 };
enum { H_PART_2Nx2N, H_PART_2NxN, H_PART_Nx2N, H_PART_NxN, H_PART_2NxnU, H_PART_2NxnD, H_PART_nLx2N, H_PART_nRx2N };
enum { HST_CU, HST_INTRA_CU, HST_SKIP_CU, HST_MERGE_PU, HST_AMVP_PU, HST_BI_PU, HST_AMP, HST_NXN, HST_TU4, HST_TU8, HST_TU16, HST_TU32, HST_DST,
       HST_SDH, HST_TSKIP, HST_BYPASS, HST_PCM, HST_DQP, HST_SAO_BAND, HST_SAO_EDGE, HST_WP, HST_TMVP, HST_SCALING, HST_WPP_ROWS, HST_TILES,
       HST_DEP_SLICE, HST_LT_REF, HST_RPLM, HST_STRONG_INTRA, HST_CIP, HST_SLICES, HST_I, HST_P, HST_B, HST_MERGE_VS_PRUNED_B1, HST_DEP_OPENS_TILE, HST_N };

typedef struct {                 /* short-term reference picture set (7.4.8) */
    int n_neg, n_pos;
    int dpoc[2][16];             /* [0] = S0 (negative), [1] = S1 (positive) */
    uint8_t used[2][16];
} StRps;

typedef struct {                 /* scaling factors m[x][y] (7.4.5), [sizeId][matrixId][y * n + x] */
    uint8_t f4[6][16], f8[6][64], f16[6][256], f32[2][1024];
} ScalingFactors;

typedef struct {
    int valid;
    int chroma_format_idc, width, height, conf_win[4] /* l r t b, chroma units */, bit_depth;
    int log2_max_poc_lsb, max_sub_layers;
    int max_dec_pic_buffering[8], max_num_reorder[8], max_latency_increase[8];
    int log2_min_cb, log2_ctb, log2_min_tb, log2_max_tb, max_th_depth_inter, max_th_depth_intra;
    int scaling_list_enabled, sps_scaling_present; ScalingFactors sf;
    int amp, sao, pcm, pcm_bits_y, pcm_bits_c, log2_min_pcm, log2_max_pcm, pcm_loop_filter_disabled;
    int n_st_rps; StRps st_rps[65];
    int long_term_present, n_lt_sps; int lt_poc_lsb[32]; uint8_t lt_used[32];
    int temporal_mvp, strong_intra_smoothing;
} HSps;

typedef struct {
    int valid, sps_id;
    int dependent_slices, output_flag_present, n_extra_bits, sign_hiding, cabac_init_present;
    int n_ref_default[2], init_qp, constrained_intra, transform_skip, cu_qp_delta, diff_cu_qp_delta_depth;
    int cb_qp_offset, cr_qp_offset, slice_chroma_qp_offsets, weighted_pred, weighted_bipred, tq_bypass;
    int tiles, wpp, n_tile_cols, n_tile_rows, uniform_spacing, col_w[20], row_h[22], lf_across_tiles;
    int lf_across_slices, deblock_control, deblock_override, deblock_disabled, beta_offset_div2, tc_offset_div2;
    int scaling_present; ScalingFactors sf;
    int lists_modification, log2_par_mrg_level, sh_extension;
} HPps;

typedef struct {
    int first_in_pic, no_output_of_prior, pps_id, dependent, segment_addr;
    int type, pic_output, poc_lsb;
    int st_rps_sps_flag, st_rps_idx; StRps st_rps;           /* the RPS in force (copied from the SPS or parsed) */
    int n_lt; int lt_poc[32]; uint8_t lt_used[32], lt_msb_present[32];   /* lt_poc: full POC when msb present, else lsb */
    int temporal_mvp, sao_luma, sao_chroma;
    int n_ref[2]; int rplm_flag[2]; int list_entry[2][16];
    int mvd_l1_zero, cabac_init_flag, collocated_from_l0, collocated_ref_idx;
    int wp_log2_denom_l, wp_log2_denom_c; int16_t wp_w[2][16][3], wp_o[2][16][3]; int has_wp;
    int max_merge_cand, qp_delta, cb_qp_offset, cr_qp_offset;
    int deblock_disabled, beta_offset_div2, tc_offset_div2, lf_across_slices;
    int n_entry;
    size_t data_offset;        /* byte offset of slice_segment_data() in the RBSP */
    /* derived */
    int slice_addr;            /* SliceAddrRs */
    int slice_qp;
    int ref_poc[2][16]; uint8_t ref_is_lt[2][16]; int8_t ref_dpb[2][16];   /* RefPicList0/1 -> POC, long-term?, DPB index */
} HSlice;

typedef struct { int16_t mv[2][2]; int8_t ref_idx[2]; uint8_t pred_flag; } HMotion;      /* pred_flag bit0 = L0, bit1 = L1 */

typedef struct HPic {
    int in_use, is_ref /* 0 no, 1 short, 2 long */, needed_for_output, pic_output, poc, decode_index, slice_type, pic_latency;
    uint8_t *pl[3]; int stride[3];
    /* collocated motion (8.5.3.2.8): per 16x16 block */
    HMotion *col_mv; int *col_ref_poc /* [2] per entry */; uint8_t *col_ref_lt, *col_intra;
    int col_w, col_h;
} HPic;

typedef struct { uint8_t type[3], band_pos[3], eo_class[3]; int8_t off[3][4]; } HSao;

struct OrchDec {
    orch_frame_cb cb; void *user;
    char err[256];
    HSps sps[16]; HPps pps[64];
    const HSps *asps; const HPps *apps;
    HPic dpb[H_MAX_DPB]; HPic *cur;
    int decode_count, poc_tid0, first_picture, no_rasl_output, seen_eos;
    int pic_started;
    /* active picture geometry */
    int w, h, ctb_w, ctb_h, ctb_size, min_cb_w, min_cb_h, w4, h4;     /* w4/h4: picture size in 4x4 units */
    /* per-picture maps at 4x4 granularity unless noted */
    uint8_t *pred_mode;        /* 0 not yet decoded, 1 inter, 2 intra */
    uint8_t *skip_flag, *ct_depth, *ipm /* IntraPredModeY */, *nofilter /* pcm+loop-filter-off or transquant bypass */;
    int8_t *qp_y;
    uint8_t *edge;             /* bit0: left edge of this 4x4 is a TU edge, bit1: top TU edge, bit2: left PU edge, bit3: top PU edge */
    uint8_t *cbf;              /* luma TU containing this 4x4 has coefficients */
    HMotion *mot;
    int16_t *slice_of4;        /* slice-table index per 4x4 (for ref POC lookup in deblocking) */
    int *ctb_slice_addr;       /* per CTB: SliceAddrRs, -1 = not decoded */
    int16_t *ctb_slice_idx;    /* per CTB: index into slices[] */
    HSao *sao;                 /* per CTB */
    int *ctb_rs2ts, *ctb_ts2rs, *tile_id /* by ts */, *col_bd, *row_bd;
    uint32_t *min_tb_zs;       /* MinTbAddrZs, [y * tbw + x] */
    int tb_w, tb_h;
    HSlice slices[H_MAX_SLICES]; int n_slices;
    uint8_t *deblocked[3];     /* scratch planes for SAO input */
    int last_cu_qp;            /* QpY of the last coding unit decoded (qPY_PREV of the next quantization group, 8.6.1) */
    uint8_t dep_st[ORCH_N_CTX], dep_mps[ORCH_N_CTX], wpp_st[ORCH_N_CTX], wpp_mps[ORCH_N_CTX]; int dep_valid, wpp_valid_pic;   /* 9.3.2.2 storage */
    /* digest / stats */
    int digest_on; uint64_t digest, digest_n;
    long stats[HST_N];
};

/* orc_hevc_ps.c */
int  orch_parse_sps(OrchDec *d, Bits *b);
int  orch_parse_pps(OrchDec *d, Bits *b);
int  orch_parse_slice_header(OrchDec *d, Bits *b, int nal_type, HSlice *sh, const HSlice *prev_independent);
void orch_default_scaling(ScalingFactors *sf);
/* orc_hevc_ctu.c */
int  orch_decode_slice_data(OrchDec *d, HSlice *sh, int slice_idx, const uint8_t *rbsp, size_t len);
/* orc_hevc_filter.c */
void orch_deblock_picture(OrchDec *d);
void orch_sao_picture(OrchDec *d);

#define H_FAIL(d, ...) do { snprintf((d)->err, sizeof (d)->err, __VA_ARGS__); return -1; } while (0)
static inline int h_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int h_clip1(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static inline int h_ceil_log2(int v) { int n = 0; while ((1 << n) < v) n++; return n; }
#endif
