/*
 * oracle/orc_parse.c -- CPU ORACLE (test infrastructure only).
 * SPS / PPS / slice-header syntax, H.264 clauses 7.3.2.1, 7.3.2.2, 7.3.3.
 * Restates the header-parsing half of the closed cuvidParseVideoData call
 * made at /root/reference/nv_dec/nv_dec.cpp:394 (parser created :278-366).
 */
#include "orc_internal.h"

static const uint8_t def4_intra[16] = {6,13,13,20,20,20,28,28,28,28,32,32,32,37,37,42};
static const uint8_t def4_inter[16] = {10,14,14,20,20,20,24,24,24,24,27,27,27,30,30,34};
static const uint8_t def8_intra[64] = {
  6,10,10,13,11,13,16,16,16,16,18,18,18,18,18,23,23,23,23,23,23,25,25,25,25,25,25,25,27,27,27,27,
  27,27,27,27,29,29,29,29,29,29,29,31,31,31,31,31,31,33,33,33,33,33,36,36,36,36,38,38,38,40,40,42};
static const uint8_t def8_inter[64] = {
  9,13,13,15,13,15,17,17,17,17,19,19,19,19,19,21,21,21,21,21,21,22,22,22,22,22,22,22,24,24,24,24,
  24,24,24,24,25,25,25,25,25,25,25,27,27,27,27,27,27,28,28,28,28,28,30,30,30,30,32,32,32,33,33,35};

/* 7.3.2.1.1.1 scaling_list(); values kept in transmitted (zig-zag) order */
static void parse_scaling_list(Bits *b, uint8_t *list, int size, int *use_default) {
    int last = 8, next = 8;
    *use_default = 0;
    for (int j = 0; j < size; j++) {
        if (next != 0) {
            int delta = bits_se(b);
            next = (last + delta + 256) % 256;
            if (j == 0 && next == 0) *use_default = 1;
        }
        list[j] = (uint8_t)(next == 0 ? last : next);
        last = list[j];
    }
}

/* parse the 8 (4:2:0) lists with fall-back rule A (defaults) or B (fb4/fb8 = SPS lists) */
static void parse_scaling_matrix(Bits *b, uint8_t s4[6][16], uint8_t s8[2][64], int n_lists,
                                 const uint8_t fb4[6][16], const uint8_t fb8[2][64]) {
    for (int i = 0; i < n_lists; i++) {
        int present = bits_u1(b), use_def = 0;
        if (i < 6) {
            if (present) parse_scaling_list(b, s4[i], 16, &use_def);
            if (!present) {                         /* fall-back */
                if (i == 0) memcpy(s4[0], fb4 ? fb4[0] : def4_intra, 16);
                else if (i == 3) memcpy(s4[3], fb4 ? fb4[3] : def4_inter, 16);
                else memcpy(s4[i], s4[i - 1], 16);
            } else if (use_def) memcpy(s4[i], i < 3 ? def4_intra : def4_inter, 16);
        } else {
            int k = i - 6;
            if (present) parse_scaling_list(b, s8[k], 64, &use_def);
            if (!present) memcpy(s8[k], fb8 ? fb8[k] : (k == 0 ? def8_intra : def8_inter), 64);
            else if (use_def) memcpy(s8[k], k == 0 ? def8_intra : def8_inter, 64);
        }
    }
}

static void skip_hrd(Bits *b) {
    int cnt = bits_ue(b) + 1;
    bits_u(b, 4); bits_u(b, 4);
    for (int i = 0; i < cnt && !b->err; i++) { bits_ue(b); bits_ue(b); bits_u1(b); }
    bits_u(b, 5); bits_u(b, 5); bits_u(b, 5); bits_u(b, 5);
}

int orc_parse_sps(OrcDec *d, Bits *b) {
    Sps s; memset(&s, 0, sizeof s);
    s.profile_idc = bits_u(b, 8);
    s.constraint_flags = bits_u(b, 8);
    s.level_idc = bits_u(b, 8);
    s.sps_id = bits_ue(b);
    if (s.sps_id > 31) ORC_FAIL(d, "sps_id out of range");
    s.chroma_format_idc = 1; s.bit_depth_luma = 8; s.bit_depth_chroma = 8;
    memset(s.scaling4, 16, sizeof s.scaling4); memset(s.scaling8, 16, sizeof s.scaling8);
    int p = s.profile_idc;
    if (p == 100 || p == 110 || p == 122 || p == 244 || p == 44 || p == 83 || p == 86 ||
        p == 118 || p == 128 || p == 138 || p == 139 || p == 134 || p == 135) {
        s.chroma_format_idc = bits_ue(b);
        if (s.chroma_format_idc == 3) bits_u1(b);
        s.bit_depth_luma = 8 + bits_ue(b);
        s.bit_depth_chroma = 8 + bits_ue(b);
        s.qpprime_y_zero_transform_bypass = bits_u1(b);
        s.scaling_matrix_present = bits_u1(b);
        if (s.scaling_matrix_present)
            parse_scaling_matrix(b, s.scaling4, s.scaling8, s.chroma_format_idc != 3 ? 8 : 12, NULL, NULL);
    }
    s.log2_max_frame_num = 4 + bits_ue(b);
    s.poc_type = bits_ue(b);
    if (s.poc_type == 0) s.log2_max_poc_lsb = 4 + bits_ue(b);
    else if (s.poc_type == 1) {
        s.delta_pic_order_always_zero = bits_u1(b);
        s.offset_for_non_ref_pic = bits_se(b);
        s.offset_for_top_to_bottom = bits_se(b);
        s.num_ref_frames_in_poc_cycle = bits_ue(b);
        if (s.num_ref_frames_in_poc_cycle > 255) ORC_FAIL(d, "bad poc cycle");
        for (int i = 0; i < s.num_ref_frames_in_poc_cycle; i++) s.offset_for_ref_frame[i] = bits_se(b);
    } else if (s.poc_type != 2) ORC_FAIL(d, "bad pic_order_cnt_type");
    s.max_num_ref_frames = bits_ue(b);
    s.gaps_in_frame_num_allowed = bits_u1(b);
    s.mb_width = bits_ue(b) + 1;
    s.mb_height = bits_ue(b) + 1;
    s.frame_mbs_only = bits_u1(b);
    if (!s.frame_mbs_only) { s.mb_aff = bits_u1(b); s.mb_height *= 2; }
    s.direct_8x8_inference = bits_u1(b);
    s.crop = bits_u1(b);
    if (s.crop) {
        s.crop_left = bits_ue(b); s.crop_right = bits_ue(b);
        s.crop_top = bits_ue(b); s.crop_bottom = bits_ue(b);
        if (!s.frame_mbs_only) { s.crop_top *= 2; s.crop_bottom *= 2; }     /* CropUnitY = SubHeightC * (2 - frame_mbs_only_flag), 7.4.2.1.1 */
    }
    s.vui_present = bits_u1(b);
    s.max_num_reorder_frames = -1; s.max_dec_frame_buffering = -1;
    if (s.vui_present) {                      /* E.1.1 vui_parameters() */
        if (bits_u1(b)) { if (bits_u(b, 8) == 255) { bits_u(b, 16); bits_u(b, 16); } }
        if (bits_u1(b)) bits_u1(b);
        if (bits_u1(b)) { bits_u(b, 3); bits_u1(b); if (bits_u1(b)) { bits_u(b, 8); bits_u(b, 8); bits_u(b, 8); } }
        if (bits_u1(b)) { bits_ue(b); bits_ue(b); }
        if (bits_u1(b)) { bits_u(b, 32); bits_u(b, 32); bits_u1(b); }
        int nal_hrd = bits_u1(b); if (nal_hrd) skip_hrd(b);
        int vcl_hrd = bits_u1(b); if (vcl_hrd) skip_hrd(b);
        if (nal_hrd || vcl_hrd) bits_u1(b);
        bits_u1(b);                           /* pic_struct_present_flag */
        s.bitstream_restriction = bits_u1(b);
        if (s.bitstream_restriction) {
            bits_u1(b); bits_ue(b); bits_ue(b); bits_ue(b); bits_ue(b);
            s.max_num_reorder_frames = bits_ue(b);
            s.max_dec_frame_buffering = bits_ue(b);
        }
    }
    if (b->err) ORC_FAIL(d, "SPS truncated");
    if (s.chroma_format_idc != 1 || s.bit_depth_luma != 8 || s.bit_depth_chroma != 8)
        ORC_FAIL(d, "unsupported chroma format / bit depth (8-bit 4:2:0 only)");
    /* frame_mbs_only_flag = 0 without MBAFF: every picture is a frame or a field picture (PAFF) */
    if (!s.frame_mbs_only && s.mb_aff) ORC_FAIL(d, "interlaced streams with MBAFF unsupported");
    if (!s.frame_mbs_only && !s.direct_8x8_inference) ORC_FAIL(d, "direct_8x8_inference_flag must be 1 when frame_mbs_only_flag is 0");
    if (s.mb_width > 1024 || s.mb_height > 1024) ORC_FAIL(d, "picture too large");
    s.valid = 1;
    d->sps[s.sps_id] = s;
    return 0;
}

int orc_parse_pps(OrcDec *d, Bits *b) {
    Pps p; memset(&p, 0, sizeof p);
    p.pps_id = bits_ue(b);
    p.sps_id = bits_ue(b);
    if (p.pps_id > 255 || p.sps_id > 31) ORC_FAIL(d, "pps/sps id out of range");
    p.entropy_coding_mode = bits_u1(b);
    p.bottom_field_pic_order_present = bits_u1(b);
    p.num_slice_groups = bits_ue(b) + 1;
    if (p.num_slice_groups > 1) ORC_FAIL(d, "FMO (slice groups) unsupported");
    p.num_ref_idx_l0_default = bits_ue(b) + 1;
    p.num_ref_idx_l1_default = bits_ue(b) + 1;
    p.weighted_pred = bits_u1(b);
    p.weighted_bipred_idc = bits_u(b, 2);
    p.pic_init_qp = 26 + bits_se(b);
    p.pic_init_qs = 26 + bits_se(b);
    p.chroma_qp_index_offset = bits_se(b);
    p.deblocking_filter_control_present = bits_u1(b);
    p.constrained_intra_pred = bits_u1(b);
    p.redundant_pic_cnt_present = bits_u1(b);
    p.second_chroma_qp_index_offset = p.chroma_qp_index_offset;
    const Sps *s = d->sps[p.sps_id].valid ? &d->sps[p.sps_id] : NULL;
    if (s) { memcpy(p.scaling4, s->scaling4, sizeof p.scaling4); memcpy(p.scaling8, s->scaling8, sizeof p.scaling8); }
    else { memset(p.scaling4, 16, sizeof p.scaling4); memset(p.scaling8, 16, sizeof p.scaling8); }
    if (bits_more_rbsp(b)) {
        p.transform_8x8_mode = bits_u1(b);
        p.scaling_matrix_present = bits_u1(b);
        if (p.scaling_matrix_present) {
            int sps_has = s && s->scaling_matrix_present;
            parse_scaling_matrix(b, p.scaling4, p.scaling8, 6 + 2 * p.transform_8x8_mode,
                                 sps_has ? s->scaling4 : NULL, sps_has ? s->scaling8 : NULL);
        }
        p.second_chroma_qp_index_offset = bits_se(b);
    }
    if (b->err) ORC_FAIL(d, "PPS truncated");
    p.valid = 1;
    d->pps[p.pps_id] = p;
    return 0;
}

int orc_parse_slice_header(OrcDec *d, Bits *b, int nal_unit_type, int nal_ref_idc, SliceHdr *sh) {
    memset(sh, 0, sizeof *sh);
    sh->nal_ref_idc = nal_ref_idc;
    sh->idr = nal_unit_type == 5;
    sh->first_mb = bits_ue(b);
    int st = bits_ue(b);
    if (st > 9) ORC_FAIL(d, "bad slice_type");
    sh->slice_type = st % 5;
    if (sh->slice_type == SLICE_SP || sh->slice_type == SLICE_SI) ORC_FAIL(d, "SP/SI slices unsupported");
    sh->pps_id = bits_ue(b);
    if (sh->pps_id > 255 || !d->pps[sh->pps_id].valid) ORC_FAIL(d, "slice refers to missing PPS");
    const Pps *pps = &d->pps[sh->pps_id];
    if (!d->sps[pps->sps_id].valid) ORC_FAIL(d, "slice refers to missing SPS");
    const Sps *sps = &d->sps[pps->sps_id];
    sh->frame_num = bits_u(b, sps->log2_max_frame_num);
    if (!sps->frame_mbs_only) { sh->field_pic = bits_u1(b); if (sh->field_pic) sh->bottom_field = bits_u1(b); }     /* 7.3.3: field_pic_flag, bottom_field_flag */
    if (sh->idr) sh->idr_pic_id = bits_ue(b);
    /* the counts of the OTHER field (delta_pic_order_cnt_bottom, delta_pic_order_cnt[1]) are sent with frames only */
    if (sps->poc_type == 0) {
        sh->poc_lsb = bits_u(b, sps->log2_max_poc_lsb);
        if (pps->bottom_field_pic_order_present && !sh->field_pic) sh->delta_poc_bottom = bits_se(b);
    } else if (sps->poc_type == 1 && !sps->delta_pic_order_always_zero) {
        sh->delta_poc[0] = bits_se(b);
        if (pps->bottom_field_pic_order_present && !sh->field_pic) sh->delta_poc[1] = bits_se(b);
    }
    if (pps->redundant_pic_cnt_present) sh->redundant_pic_cnt = bits_ue(b);
    if (sh->slice_type == SLICE_B) sh->direct_spatial_mv_pred = bits_u1(b);
    sh->num_ref_idx[0] = pps->num_ref_idx_l0_default;
    sh->num_ref_idx[1] = pps->num_ref_idx_l1_default;
    if (sh->slice_type == SLICE_P || sh->slice_type == SLICE_B) {
        if (bits_u1(b)) {
            sh->num_ref_idx[0] = bits_ue(b) + 1;
            if (sh->slice_type == SLICE_B) sh->num_ref_idx[1] = bits_ue(b) + 1;
        }
        /* 7.4.3: at most 16 entries in a frame's list, 32 in a field's */
        if (sh->num_ref_idx[0] > (sh->field_pic ? 32 : 16) || sh->num_ref_idx[1] > (sh->field_pic ? 32 : 16)) ORC_FAIL(d, "num_ref_idx out of range");
    }
    if (sh->slice_type != SLICE_B) sh->num_ref_idx[1] = 0;
    if (sh->slice_type == SLICE_I) sh->num_ref_idx[0] = 0;
    /* 7.3.3.1 ref_pic_list_modification() */
    int nlists = sh->slice_type == SLICE_I ? 0 : (sh->slice_type == SLICE_B ? 2 : 1);
    for (int l = 0; l < nlists; l++) {
        sh->rplm_flag[l] = bits_u1(b);
        if (sh->rplm_flag[l]) {
            for (;;) {
                int idc = bits_ue(b);
                if (idc == 3) break;
                if (idc > 3 || sh->n_rplm[l] >= 66 || b->err) ORC_FAIL(d, "bad ref_pic_list_modification");
                sh->rplm[l][sh->n_rplm[l]].idc = idc;
                sh->rplm[l][sh->n_rplm[l]].val = bits_ue(b);
                sh->n_rplm[l]++;
            }
        }
    }
    /* 7.3.3.2 pred_weight_table() */
    if ((pps->weighted_pred && sh->slice_type == SLICE_P) ||
        (pps->weighted_bipred_idc == 1 && sh->slice_type == SLICE_B)) {
        sh->luma_log2_wd = bits_ue(b);
        sh->chroma_log2_wd = bits_ue(b);
        if (sh->luma_log2_wd > 7 || sh->chroma_log2_wd > 7) ORC_FAIL(d, "bad weight denominator");
        for (int l = 0; l < nlists; l++)
            for (int i = 0; i < sh->num_ref_idx[l]; i++) {
                sh->luma_weight[l][i] = 1 << sh->luma_log2_wd;
                sh->chroma_weight[l][i][0] = sh->chroma_weight[l][i][1] = 1 << sh->chroma_log2_wd;
                sh->luma_weight_flag[l][i] = bits_u1(b);
                if (sh->luma_weight_flag[l][i]) { sh->luma_weight[l][i] = bits_se(b); sh->luma_offset[l][i] = bits_se(b); }
                sh->chroma_weight_flag[l][i] = bits_u1(b);
                if (sh->chroma_weight_flag[l][i])
                    for (int j = 0; j < 2; j++) { sh->chroma_weight[l][i][j] = bits_se(b); sh->chroma_offset[l][i][j] = bits_se(b); }
            }
    }
    /* 7.3.3.3 dec_ref_pic_marking() */
    if (nal_ref_idc != 0) {
        if (sh->idr) { sh->no_output_of_prior_pics = bits_u1(b); sh->long_term_reference_flag = bits_u1(b); }
        else {
            sh->adaptive_marking = bits_u1(b);
            if (sh->adaptive_marking) {
                for (;;) {
                    int op = bits_ue(b);
                    if (op == 0) break;
                    if (op > 6 || sh->n_mmco >= 66 || b->err) ORC_FAIL(d, "bad MMCO");
                    Mmco *m = &sh->mmco[sh->n_mmco++];
                    memset(m, 0, sizeof *m); m->op = op;
                    if (op == 1 || op == 3) m->diff_pic_nums_minus1 = bits_ue(b);
                    if (op == 2) m->long_term_pic_num = bits_ue(b);
                    if (op == 3 || op == 6) m->long_term_frame_idx = bits_ue(b);
                    if (op == 4) m->max_long_term_frame_idx_plus1 = bits_ue(b);
                }
            }
        }
    }
    if (pps->entropy_coding_mode && sh->slice_type != SLICE_I) sh->cabac_init_idc = bits_ue(b);
    if (sh->cabac_init_idc > 2) ORC_FAIL(d, "bad cabac_init_idc");
    sh->slice_qp_delta = bits_se(b);
    sh->qp = pps->pic_init_qp + sh->slice_qp_delta;
    if (sh->qp < 0 || sh->qp > 51) ORC_FAIL(d, "slice QP out of range");
    if (pps->deblocking_filter_control_present) {
        sh->disable_deblock = bits_ue(b);
        if (sh->disable_deblock > 2) ORC_FAIL(d, "bad disable_deblocking_filter_idc");
        if (sh->disable_deblock != 1) { sh->alpha_c0_offset = 2 * bits_se(b); sh->beta_offset = 2 * bits_se(b); }
    }
    if (b->err) ORC_FAIL(d, "slice header truncated");
    return 0;
}
