/*
 * oracle/orc_hevc_ps.c -- CPU ORACLE (test infrastructure only): HEVC parameter sets and slice segment header
 * (ITU-T H.265 7.3.2.2, 7.3.2.3, 7.3.3, 7.3.4, 7.3.6, 7.3.7 with the semantics of 7.4).  Restates the part of
 * cuvidParseVideoData that fills CUVIDHEVCPICPARAMS (/root/reference/nv_sdk/inc/dynlink_cuviddec.h:428-530).
 */
#include "orc_hevc_internal.h"

static void diag_scan(int n, uint8_t *sx, uint8_t *sy) {      /* 6.5.3 up-right diagonal */
    int i = 0, x = 0, y = 0;
    for (;;) {
        while (y >= 0) { if (x < n && y < n) { sx[i] = (uint8_t)x; sy[i] = (uint8_t)y; i++; } y--; x++; }
        y = x; x = 0;
        if (i >= n * n) break;
    }
}

/* expand coded lists (diagonal order) into m[x][y]; lists: l4[6][16], l8[6][64], l16[6][64] + dc16[6], l32[2][64] + dc32[2] */
typedef struct { uint8_t l4[6][16], l8[6][64], l16[6][64], l32[6][64]; uint8_t dc16[6], dc32[6]; } CodedLists;
static void expand_lists(const CodedLists *c, ScalingFactors *sf) {
    uint8_t x4[16], y4[16], x8[64], y8[64];
    diag_scan(4, x4, y4); diag_scan(8, x8, y8);
    for (int m = 0; m < 6; m++) {
        for (int i = 0; i < 16; i++) sf->f4[m][y4[i] * 4 + x4[i]] = c->l4[m][i];
        for (int i = 0; i < 64; i++) sf->f8[m][y8[i] * 8 + x8[i]] = c->l8[m][i];
        for (int i = 0; i < 64; i++)
            for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++) sf->f16[m][(y8[i] * 2 + j) * 16 + x8[i] * 2 + k] = c->l16[m][i];
        sf->f16[m][0] = c->dc16[m];
    }
    for (int m = 0; m < 2; m++) {
        for (int i = 0; i < 64; i++)
            for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) sf->f32[m][(y8[i] * 4 + j) * 32 + x8[i] * 4 + k] = c->l32[m * 3][i];
        sf->f32[m][0] = c->dc32[m * 3];
    }
}
static void default_lists(CodedLists *c) {
    memset(c->l4, 16, sizeof c->l4);
    for (int m = 0; m < 6; m++) {
        memcpy(c->l8[m], orch_scaling_default[m >= 3], 64); memcpy(c->l16[m], orch_scaling_default[m >= 3], 64);
        memcpy(c->l32[m], orch_scaling_default[m >= 3], 64);
        c->dc16[m] = c->dc32[m] = 16;
    }
}
void orch_default_scaling(ScalingFactors *sf) { CodedLists c; default_lists(&c); expand_lists(&c, sf); }

/* 7.3.4 scaling_list_data() */
static int parse_scaling_list_data(Bits *b, ScalingFactors *sf) {
    CodedLists c; default_lists(&c);
    for (int size_id = 0; size_id < 4; size_id++) {
        int n_mat = size_id == 3 ? 2 : 6;
        for (int k = 0; k < n_mat; k++) {
            int m = size_id == 3 ? k * 3 : k;
            uint8_t *list = size_id == 0 ? c.l4[m] : size_id == 1 ? c.l8[m] : size_id == 2 ? c.l16[m] : c.l32[m];
            uint8_t *dc = size_id == 2 ? &c.dc16[m] : size_id == 3 ? &c.dc32[m] : NULL;
            int n = size_id == 0 ? 16 : 64;
            if (!bits_u1(b)) {                                  /* scaling_list_pred_mode_flag == 0 */
                unsigned delta = bits_ue(b);
                if (delta > (unsigned)k) return -1;
                if (delta == 0) {                               /* default list */
                    if (size_id == 0) memset(list, 16, 16); else memcpy(list, orch_scaling_default[size_id == 3 ? k : (m >= 3)], 64);
                    if (dc) *dc = 16;
                } else {
                    int rm = size_id == 3 ? (k - (int)delta) * 3 : m - (int)delta;
                    const uint8_t *ref = size_id == 0 ? c.l4[rm] : size_id == 1 ? c.l8[rm] : size_id == 2 ? c.l16[rm] : c.l32[rm];
                    memmove(list, ref, (size_t)n);
                    if (dc) *dc = size_id == 2 ? c.dc16[rm] : c.dc32[rm];
                }
            } else {
                int next = 8;
                if (size_id > 1) { int v = bits_se(b); if (v < -7 || v > 247) return -1; next = v + 8; *dc = (uint8_t)next; }
                for (int i = 0; i < n; i++) { int dl = bits_se(b); if (dl < -128 || dl > 127) return -1; next = (next + dl + 256) % 256;
                    list[i] = (uint8_t)next; }
            }
        }
    }
    if (b->err) return -1;
    expand_lists(&c, sf);
    return 0;
}

/* 7.3.3 profile_tier_level(1, maxNumSubLayersMinus1) */
static void skip_ptl(Bits *b, int max_sub_minus1) {
    bits_u(b, 8); bits_u(b, 32); bits_u(b, 4); bits_u(b, 32); bits_u(b, 12); bits_u(b, 8);     /* 2+1+5, 32 compat, 4 flags, 43+1 reserved, level */
    int prof[8] = {0}, lev[8] = {0};
    for (int i = 0; i < max_sub_minus1; i++) { prof[i] = (int)bits_u1(b); lev[i] = (int)bits_u1(b); }
    if (max_sub_minus1 > 0) for (int i = max_sub_minus1; i < 8; i++) bits_u(b, 2);
    for (int i = 0; i < max_sub_minus1; i++) {
        if (prof[i]) { bits_u(b, 32); bits_u(b, 32); bits_u(b, 24); }
        if (lev[i]) bits_u(b, 8);
    }
}

/* 7.3.7 st_ref_pic_set(idx); `sets` holds the already decoded sets 0 .. idx-1 (n_sps of them live in the SPS) */
static int parse_st_rps(Bits *b, StRps *out, int idx, int n_sps, const StRps *sets) {
    memset(out, 0, sizeof *out);
    int inter = idx != 0 ? (int)bits_u1(b) : 0;
    if (inter) {
        int delta_idx = 1;
        if (idx == n_sps) delta_idx = (int)bits_ue(b) + 1;
        if (delta_idx > idx) return -1;
        const StRps *ref = &sets[idx - delta_idx];
        int sign = (int)bits_u1(b), absd = (int)bits_ue(b) + 1;
        int drps = (1 - 2 * sign) * absd;
        int n_ref = ref->n_neg + ref->n_pos;
        uint8_t used[33], use_delta[33];
        for (int j = 0; j <= n_ref; j++) { used[j] = (uint8_t)bits_u1(b); use_delta[j] = 1; if (!used[j]) use_delta[j] = (uint8_t)bits_u1(b); }
        int i = 0;
        for (int j = ref->n_pos - 1; j >= 0; j--) { int dp = ref->dpoc[1][j] + drps; if (dp < 0 && use_delta[ref->n_neg + j]) { if (i >= 16) return -1;
            out->dpoc[0][i] = dp; out->used[0][i++] = used[ref->n_neg + j]; } }
        if (drps < 0 && use_delta[n_ref]) { if (i >= 16) return -1; out->dpoc[0][i] = drps; out->used[0][i++] = used[n_ref]; }
        for (int j = 0; j < ref->n_neg; j++) { int dp = ref->dpoc[0][j] + drps; if (dp < 0 && use_delta[j]) { if (i >= 16) return -1; out->dpoc[0][i] = dp;
            out->used[0][i++] = used[j]; } }
        out->n_neg = i;
        i = 0;
        for (int j = ref->n_neg - 1; j >= 0; j--) { int dp = ref->dpoc[0][j] + drps; if (dp > 0 && use_delta[j]) { if (i >= 16) return -1;
            out->dpoc[1][i] = dp; out->used[1][i++] = used[j]; } }
        if (drps > 0 && use_delta[n_ref]) { if (i >= 16) return -1; out->dpoc[1][i] = drps; out->used[1][i++] = used[n_ref]; }
        for (int j = 0; j < ref->n_pos; j++) { int dp = ref->dpoc[1][j] + drps; if (dp > 0 && use_delta[ref->n_neg + j]) { if (i >= 16) return -1;
            out->dpoc[1][i] = dp; out->used[1][i++] = used[ref->n_neg + j]; } }
        out->n_pos = i;
    } else {
        unsigned nn = bits_ue(b), np = bits_ue(b);
        if (nn > 16 || np > 16 || nn + np > 16) return -1;
        out->n_neg = (int)nn; out->n_pos = (int)np;
        int prev = 0;
        for (int i = 0; i < out->n_neg; i++) { prev -= (int)bits_ue(b) + 1; out->dpoc[0][i] = prev; out->used[0][i] = (uint8_t)bits_u1(b); }
        prev = 0;
        for (int i = 0; i < out->n_pos; i++) { prev += (int)bits_ue(b) + 1; out->dpoc[1][i] = prev; out->used[1][i] = (uint8_t)bits_u1(b); }
    }
    return b->err ? -1 : 0;
}

int orch_parse_sps(OrchDec *d, Bits *b) {
    HSps s; memset(&s, 0, sizeof s);
    bits_u(b, 4);
    s.max_sub_layers = (int)bits_u(b, 3) + 1;
    bits_u1(b);
    if (s.max_sub_layers > 7) H_FAIL(d, "SPS: sps_max_sub_layers_minus1 out of range");
    skip_ptl(b, s.max_sub_layers - 1);
    unsigned id = bits_ue(b);
    if (id > 15) H_FAIL(d, "SPS id out of range");
    s.chroma_format_idc = (int)bits_ue(b);
    if (s.chroma_format_idc != 1) H_FAIL(d, "unsupported chroma format (4:2:0 only)");
    s.width = (int)bits_ue(b); s.height = (int)bits_ue(b);
    if (bits_u1(b)) for (int i = 0; i < 4; i++) s.conf_win[i] = (int)bits_ue(b);
    int bdl = (int)bits_ue(b), bdc = (int)bits_ue(b);
    if (bdl != 0 || bdc != 0) H_FAIL(d, "unsupported bit depth (8-bit only)");
    s.bit_depth = 8;
    s.log2_max_poc_lsb = (int)bits_ue(b) + 4;
    if (s.log2_max_poc_lsb > 16) H_FAIL(d, "SPS: log2_max_pic_order_cnt_lsb out of range");
    int info_present = (int)bits_u1(b);
    for (int i = info_present ? 0 : s.max_sub_layers - 1; i < s.max_sub_layers; i++) {
        s.max_dec_pic_buffering[i] = (int)bits_ue(b) + 1; s.max_num_reorder[i] = (int)bits_ue(b); s.max_latency_increase[i] = (int)bits_ue(b);
        if (s.max_dec_pic_buffering[i] > 16 || s.max_num_reorder[i] > s.max_dec_pic_buffering[i] - 1) H_FAIL(d, "SPS: DPB parameters out of range");
    }
    if (!info_present) for (int i = 0; i < s.max_sub_layers - 1; i++) { s.max_dec_pic_buffering[i] = s.max_dec_pic_buffering[s.max_sub_layers - 1];
        s.max_num_reorder[i] = s.max_num_reorder[s.max_sub_layers - 1]; s.max_latency_increase[i] = s.max_latency_increase[s.max_sub_layers - 1]; }
    s.log2_min_cb = (int)bits_ue(b) + 3;
    s.log2_ctb = s.log2_min_cb + (int)bits_ue(b);
    s.log2_min_tb = (int)bits_ue(b) + 2;
    s.log2_max_tb = s.log2_min_tb + (int)bits_ue(b);
    s.max_th_depth_inter = (int)bits_ue(b); s.max_th_depth_intra = (int)bits_ue(b);
    if (b->err || s.log2_ctb < 4 || s.log2_ctb > 6 || s.log2_min_cb > s.log2_ctb || s.log2_min_tb >= s.log2_min_cb || s.log2_max_tb > 5 ||
        s.log2_max_tb > s.log2_ctb ||
        s.max_th_depth_inter > s.log2_ctb - s.log2_min_tb || s.max_th_depth_intra > s.log2_ctb - s.log2_min_tb)
        H_FAIL(d, "SPS: block size parameters out of range");
    if (s.width <= 0 || s.height <= 0 || s.width > 8192 || s.height > 8192 || (s.width & ((1 << s.log2_min_cb) - 1)) || (s.height & ((1 << s.log2_min_cb) - 1)))
        H_FAIL(d, "SPS: picture size %dx%d not a multiple of the minimum coding block", s.width, s.height);
    if (2 * (s.conf_win[0] + s.conf_win[1]) >= s.width || 2 * (s.conf_win[2] + s.conf_win[3]) >= s.height) H_FAIL(d,
        "SPS: conformance window larger than the picture");
    s.scaling_list_enabled = (int)bits_u1(b);
    orch_default_scaling(&s.sf);
    if (s.scaling_list_enabled) { s.sps_scaling_present = (int)bits_u1(b); if (s.sps_scaling_present && parse_scaling_list_data(b, &s.sf) < 0) H_FAIL(d,
        "SPS: bad scaling_list_data"); }
    s.amp = (int)bits_u1(b); s.sao = (int)bits_u1(b); s.pcm = (int)bits_u1(b);
    if (s.pcm) {
        s.pcm_bits_y = (int)bits_u(b, 4) + 1; s.pcm_bits_c = (int)bits_u(b, 4) + 1;
        s.log2_min_pcm = (int)bits_ue(b) + 3; s.log2_max_pcm = s.log2_min_pcm + (int)bits_ue(b);
        s.pcm_loop_filter_disabled = (int)bits_u1(b);
        if (s.pcm_bits_y > 8 || s.pcm_bits_c > 8 || s.log2_min_pcm < s.log2_min_cb || s.log2_max_pcm > 5 || s.log2_max_pcm > s.log2_ctb) H_FAIL(d,
            "SPS: PCM parameters out of range");
    }
    s.n_st_rps = (int)bits_ue(b);
    if (s.n_st_rps > 64) H_FAIL(d, "SPS: too many short-term RPS");
    for (int i = 0; i < s.n_st_rps; i++) if (parse_st_rps(b, &s.st_rps[i], i, s.n_st_rps, s.st_rps) < 0) H_FAIL(d, "SPS: bad st_ref_pic_set %d", i);
    s.long_term_present = (int)bits_u1(b);
    if (s.long_term_present) {
        s.n_lt_sps = (int)bits_ue(b);
        if (s.n_lt_sps > 32) H_FAIL(d, "SPS: too many long-term pictures");
        for (int i = 0; i < s.n_lt_sps; i++) { s.lt_poc_lsb[i] = (int)bits_u(b, s.log2_max_poc_lsb); s.lt_used[i] = (uint8_t)bits_u1(b); }
    }
    s.temporal_mvp = (int)bits_u1(b); s.strong_intra_smoothing = (int)bits_u1(b);
    /* vui_parameters_present_flag and everything after it do not influence the decoding process */
    if (b->err) H_FAIL(d, "SPS truncated");
    s.valid = 1;
    if (d->asps == &d->sps[id] && d->pic_started) H_FAIL(d, "SPS %u re-sent inside a picture", id);
    d->sps[id] = s;
    return 0;
}

int orch_parse_pps(OrchDec *d, Bits *b) {
    HPps p; memset(&p, 0, sizeof p);
    unsigned id = bits_ue(b), sid = bits_ue(b);
    if (id > 63 || sid > 15) H_FAIL(d, "PPS id out of range");
    p.sps_id = (int)sid;
    p.dependent_slices = (int)bits_u1(b); p.output_flag_present = (int)bits_u1(b); p.n_extra_bits = (int)bits_u(b, 3);
    p.sign_hiding = (int)bits_u1(b); p.cabac_init_present = (int)bits_u1(b);
    p.n_ref_default[0] = (int)bits_ue(b) + 1; p.n_ref_default[1] = (int)bits_ue(b) + 1;
    p.init_qp = 26 + bits_se(b);
    p.constrained_intra = (int)bits_u1(b); p.transform_skip = (int)bits_u1(b);
    p.cu_qp_delta = (int)bits_u1(b);
    if (p.cu_qp_delta) p.diff_cu_qp_delta_depth = (int)bits_ue(b);
    p.cb_qp_offset = bits_se(b); p.cr_qp_offset = bits_se(b);
    p.slice_chroma_qp_offsets = (int)bits_u1(b); p.weighted_pred = (int)bits_u1(b); p.weighted_bipred = (int)bits_u1(b);
    p.tq_bypass = (int)bits_u1(b); p.tiles = (int)bits_u1(b); p.wpp = (int)bits_u1(b);
    p.n_tile_cols = p.n_tile_rows = 1; p.lf_across_tiles = 1; p.uniform_spacing = 1;
    if (p.tiles) {
        p.n_tile_cols = (int)bits_ue(b) + 1; p.n_tile_rows = (int)bits_ue(b) + 1;
        if (p.n_tile_cols > 20 || p.n_tile_rows > 22) H_FAIL(d, "PPS: too many tiles");
        p.uniform_spacing = (int)bits_u1(b);
        if (!p.uniform_spacing) {
            for (int i = 0; i < p.n_tile_cols - 1; i++) p.col_w[i] = (int)bits_ue(b) + 1;
            for (int i = 0; i < p.n_tile_rows - 1; i++) p.row_h[i] = (int)bits_ue(b) + 1;
        }
        p.lf_across_tiles = (int)bits_u1(b);
    }
    p.lf_across_slices = (int)bits_u1(b);
    p.deblock_control = (int)bits_u1(b);
    if (p.deblock_control) {
        p.deblock_override = (int)bits_u1(b); p.deblock_disabled = (int)bits_u1(b);
        if (!p.deblock_disabled) { p.beta_offset_div2 = bits_se(b); p.tc_offset_div2 = bits_se(b); }
    }
    p.scaling_present = (int)bits_u1(b);
    if (p.scaling_present && parse_scaling_list_data(b, &p.sf) < 0) H_FAIL(d, "PPS: bad scaling_list_data");
    p.lists_modification = (int)bits_u1(b);
    p.log2_par_mrg_level = (int)bits_ue(b) + 2;
    p.sh_extension = (int)bits_u1(b);
    if (b->err) H_FAIL(d, "PPS truncated");
    if (p.n_ref_default[0] > 15 || p.n_ref_default[1] > 15 || p.init_qp < 0 || p.init_qp > 51 || p.cb_qp_offset < -12 || p.cb_qp_offset > 12 ||
        p.cr_qp_offset < -12 || p.cr_qp_offset > 12 ||
        p.beta_offset_div2 < -6 || p.beta_offset_div2 > 6 || p.tc_offset_div2 < -6 || p.tc_offset_div2 > 6 || p.diff_cu_qp_delta_depth > 3 ||
            p.log2_par_mrg_level > 6)
        H_FAIL(d, "PPS: parameter out of range");
    p.valid = 1;
    if (d->apps == &d->pps[id] && d->pic_started) H_FAIL(d, "PPS %u re-sent inside a picture", id);
    d->pps[id] = p;
    return 0;
}

/* 7.3.6.3 pred_weight_table() */
static int parse_pwt(Bits *b, HSlice *sh) {
    sh->wp_log2_denom_l = (int)bits_ue(b);
    sh->wp_log2_denom_c = sh->wp_log2_denom_l + bits_se(b);
    if (sh->wp_log2_denom_l > 7 || sh->wp_log2_denom_c < 0 || sh->wp_log2_denom_c > 7) {
        if (getenv("ORCH_DBG")) fprintf(stderr, "pwt denom %d %d type %d nref %d %d\n", sh->wp_log2_denom_l, sh->wp_log2_denom_c, sh->type, sh->n_ref[0],
        sh->n_ref[1]); return -1; }
    for (int l = 0; l < (sh->type == H_SLICE_B ? 2 : 1); l++) {
        uint8_t lf[16], cf[16];
        for (int i = 0; i < sh->n_ref[l]; i++) lf[i] = (uint8_t)bits_u1(b);
        for (int i = 0; i < sh->n_ref[l]; i++) cf[i] = (uint8_t)bits_u1(b);
        for (int i = 0; i < sh->n_ref[l]; i++) {
            sh->wp_w[l][i][0] = (int16_t)(1 << sh->wp_log2_denom_l); sh->wp_o[l][i][0] = 0;
            sh->wp_w[l][i][1] = sh->wp_w[l][i][2] = (int16_t)(1 << sh->wp_log2_denom_c); sh->wp_o[l][i][1] = sh->wp_o[l][i][2] = 0;
            if (lf[i]) {
                int dw = bits_se(b), o = bits_se(b);
                if (dw < -128 || dw > 127 || o < -128 || o > 127) {
                    if (getenv("ORCH_DBG")) fprintf(stderr, "pwt luma l%d i%d dw %d o %d denom %d %d nref %d %d\n", l, i, dw, o, sh->wp_log2_denom_l,
                    sh->wp_log2_denom_c, sh->n_ref[0], sh->n_ref[1]); return -1; }
                sh->wp_w[l][i][0] = (int16_t)(sh->wp_w[l][i][0] + dw); sh->wp_o[l][i][0] = (int16_t)o;
            }
            if (cf[i]) for (int j = 1; j < 3; j++) {
                int dw = bits_se(b), dofs = bits_se(b);
                if (dw < -128 || dw > 127 || dofs < -512 || dofs > 511) { if (getenv("ORCH_DBG")) fprintf(stderr, "pwt chroma l%d i%d j%d dw %d dofs %d\n", l,
                    i, j, dw, dofs); return -1; }
                int w = (1 << sh->wp_log2_denom_c) + dw;
                sh->wp_w[l][i][j] = (int16_t)w;
                sh->wp_o[l][i][j] = (int16_t)h_clip3(-128, 127, (128 + dofs) - ((128 * w) >> sh->wp_log2_denom_c));
            }
        }
    }
    sh->has_wp = 1;
    if (b->err && getenv("ORCH_DBG")) fprintf(stderr, "pwt ran out of bits\n");
    return b->err ? -1 : 0;
}

/* 7.3.6.1 slice_segment_header().  `prev` = the previous independent slice segment header of this picture (dependent segments copy it). */
int orch_parse_slice_header(OrchDec *d, Bits *b, int nal_type, HSlice *sh, const HSlice *prev) {
    memset(sh, 0, sizeof *sh);
    sh->first_in_pic = (int)bits_u1(b);
    if (nal_type >= 16 && nal_type <= 23) sh->no_output_of_prior = (int)bits_u1(b);
    unsigned pid = bits_ue(b);
    if (pid > 63 || !d->pps[pid].valid) H_FAIL(d, "slice refers to missing PPS %u", pid);
    const HPps *pps = &d->pps[pid];
    if (!d->sps[pps->sps_id].valid) H_FAIL(d, "slice refers to missing SPS %d", pps->sps_id);
    const HSps *sps = &d->sps[pps->sps_id];
    int ctb = 1 << sps->log2_ctb, ctb_w = (sps->width + ctb - 1) >> sps->log2_ctb, ctb_h = (sps->height + ctb - 1) >> sps->log2_ctb;
    int dependent = 0, addr = 0;
    if (!sh->first_in_pic) {
        if (pps->dependent_slices) dependent = (int)bits_u1(b);
        addr = (int)bits_u(b, h_ceil_log2(ctb_w * ctb_h));
        if (addr >= ctb_w * ctb_h || addr == 0) H_FAIL(d, "slice_segment_address out of range");
    }
    if (dependent) {
        if (!prev) H_FAIL(d, "dependent slice segment without a preceding independent one");
        int keep_first = sh->first_in_pic, keep_noout = sh->no_output_of_prior;
        *sh = *prev;
        sh->first_in_pic = keep_first; sh->no_output_of_prior = keep_noout;
    }
    sh->pps_id = (int)pid; sh->dependent = dependent; sh->segment_addr = addr;
    if (!dependent) {
        sh->slice_addr = addr;
        bits_u(b, pps->n_extra_bits);
        sh->type = (int)bits_ue(b);
        if (sh->type > 2) H_FAIL(d, "bad slice_type");
        if (nal_type >= 16 && nal_type <= 23 && sh->type != H_SLICE_I) H_FAIL(d, "IRAP picture with a non-I slice");
        sh->pic_output = 1;
        if (pps->output_flag_present) sh->pic_output = (int)bits_u1(b);
        if (nal_type != 19 && nal_type != 20) {
            sh->poc_lsb = (int)bits_u(b, sps->log2_max_poc_lsb);
            sh->st_rps_sps_flag = (int)bits_u1(b);
            if (!sh->st_rps_sps_flag) { if (parse_st_rps(b, &sh->st_rps, sps->n_st_rps, sps->n_st_rps, sps->st_rps) < 0) H_FAIL(d,
                "bad st_ref_pic_set in slice header"); }
            else {
                if (sps->n_st_rps == 0) H_FAIL(d, "short_term_ref_pic_set_sps_flag without SPS sets");
                if (sps->n_st_rps > 1) sh->st_rps_idx = (int)bits_u(b, h_ceil_log2(sps->n_st_rps));
                if (sh->st_rps_idx >= sps->n_st_rps) H_FAIL(d, "short_term_ref_pic_set_idx out of range");
                sh->st_rps = sps->st_rps[sh->st_rps_idx];
            }
            if (sps->long_term_present) {
                int n_sps = 0;
                if (sps->n_lt_sps > 0) n_sps = (int)bits_ue(b);
                int n_pics = (int)bits_ue(b);
                if (n_sps < 0 || n_pics < 0 || n_sps > sps->n_lt_sps || n_sps + n_pics > 32) H_FAIL(d, "too many long-term pictures in slice header");
                sh->n_lt = n_sps + n_pics;
                int prev_msb = 0;
                for (int i = 0; i < sh->n_lt; i++) {
                    int lsb;
                    if (i < n_sps) { int idx = sps->n_lt_sps > 1 ? (int)bits_u(b, h_ceil_log2(sps->n_lt_sps)) : 0;
                        if (idx >= sps->n_lt_sps) H_FAIL(d, "lt_idx_sps out of range"); lsb = sps->lt_poc_lsb[idx]; sh->lt_used[i] = sps->lt_used[idx]; }
                    else { lsb = (int)bits_u(b, sps->log2_max_poc_lsb); sh->lt_used[i] = (uint8_t)bits_u1(b); }
                    sh->lt_msb_present[i] = (uint8_t)bits_u1(b);
                    int cycle = 0;
                    if (sh->lt_msb_present[i]) { cycle = (int)bits_ue(b); if (i != 0 && i != n_sps) cycle += prev_msb; prev_msb = cycle; }
                        /* DeltaPocMsbCycleLt (7-52) */
                    else if (i == 0 || i == n_sps) prev_msb = 0;
                    sh->lt_poc[i] = sh->lt_msb_present[i] ? -(cycle << sps->log2_max_poc_lsb) + lsb : lsb;
                    /* resolved against the current POC in orc_hevc_dec.c */
                }
            }
            if (sps->temporal_mvp) sh->temporal_mvp = (int)bits_u1(b);
        }
        if (sps->sao) { sh->sao_luma = (int)bits_u1(b); sh->sao_chroma = (int)bits_u1(b); }
        sh->n_ref[0] = sh->n_ref[1] = 0;
        sh->collocated_from_l0 = 1;
        sh->max_merge_cand = 5;
        if (sh->type != H_SLICE_I) {
            sh->n_ref[0] = pps->n_ref_default[0]; sh->n_ref[1] = sh->type == H_SLICE_B ? pps->n_ref_default[1] : 0;
            if (bits_u1(b)) { sh->n_ref[0] = (int)bits_ue(b) + 1; if (sh->type == H_SLICE_B) sh->n_ref[1] = (int)bits_ue(b) + 1; }
            if (sh->n_ref[0] > 15 || sh->n_ref[1] > 15) H_FAIL(d, "num_ref_idx_active out of range");
            int n_total = 0;
            for (int i = 0; i < sh->st_rps.n_neg; i++) n_total += sh->st_rps.used[0][i];
            for (int i = 0; i < sh->st_rps.n_pos; i++) n_total += sh->st_rps.used[1][i];
            for (int i = 0; i < sh->n_lt; i++) n_total += sh->lt_used[i];
            if (n_total == 0) H_FAIL(d, "P/B slice with an empty reference picture set");
            if (pps->lists_modification && n_total > 1) {
                int nb = h_ceil_log2(n_total);
                for (int l = 0; l < (sh->type == H_SLICE_B ? 2 : 1); l++) {
                    sh->rplm_flag[l] = (int)bits_u1(b);
                    if (sh->rplm_flag[l]) for (int i = 0; i < sh->n_ref[l]; i++) { sh->list_entry[l][i] = (int)bits_u(b, nb);
                        if (sh->list_entry[l][i] >= n_total) H_FAIL(d, "list_entry out of range"); }
                }
            }
            if (sh->type == H_SLICE_B) sh->mvd_l1_zero = (int)bits_u1(b);
            if (pps->cabac_init_present) sh->cabac_init_flag = (int)bits_u1(b);
            if (sh->temporal_mvp) {
                if (sh->type == H_SLICE_B) sh->collocated_from_l0 = (int)bits_u1(b);
                if ((sh->collocated_from_l0 && sh->n_ref[0] > 1) || (!sh->collocated_from_l0 && sh->n_ref[1] > 1)) sh->collocated_ref_idx = (int)bits_ue(b);
                if (sh->collocated_ref_idx >= sh->n_ref[sh->collocated_from_l0 ? 0 : 1]) H_FAIL(d, "collocated_ref_idx out of range");
            }
            if ((pps->weighted_pred && sh->type == H_SLICE_P) || (pps->weighted_bipred && sh->type == H_SLICE_B)) if (parse_pwt(b, sh) < 0) H_FAIL(d,
                "bad pred_weight_table");
            sh->max_merge_cand = 5 - (int)bits_ue(b);
            if (sh->max_merge_cand < 1 || sh->max_merge_cand > 5) H_FAIL(d, "five_minus_max_num_merge_cand out of range");
        }
        sh->qp_delta = bits_se(b);
        if (pps->slice_chroma_qp_offsets) { sh->cb_qp_offset = bits_se(b); sh->cr_qp_offset = bits_se(b); }
        if (sh->cb_qp_offset < -12 || sh->cb_qp_offset > 12 || sh->cr_qp_offset < -12 || sh->cr_qp_offset > 12) H_FAIL(d,
            "slice chroma QP offset out of range");
        int override = 0;
        if (pps->deblock_override) override = (int)bits_u1(b);
        sh->deblock_disabled = pps->deblock_disabled; sh->beta_offset_div2 = pps->beta_offset_div2; sh->tc_offset_div2 = pps->tc_offset_div2;
        if (override) {
            sh->deblock_disabled = (int)bits_u1(b);
            if (!sh->deblock_disabled) { sh->beta_offset_div2 = bits_se(b); sh->tc_offset_div2 = bits_se(b); }
            if (sh->beta_offset_div2 < -6 || sh->beta_offset_div2 > 6 || sh->tc_offset_div2 < -6 || sh->tc_offset_div2 > 6) H_FAIL(d,
                "slice deblocking offsets out of range");
        }
        sh->lf_across_slices = pps->lf_across_slices;
        if (pps->lf_across_slices && (sh->sao_luma || sh->sao_chroma || !sh->deblock_disabled)) sh->lf_across_slices = (int)bits_u1(b);
        sh->slice_qp = pps->init_qp + sh->qp_delta;
        if (sh->slice_qp < 0 || sh->slice_qp > 51) H_FAIL(d, "slice QP out of range");
    }
    sh->n_entry = 0;
    if (pps->tiles || pps->wpp) {
        unsigned n = bits_ue(b);
        if (n > (unsigned)(ctb_w * ctb_h)) H_FAIL(d, "num_entry_point_offsets out of range");
        sh->n_entry = (int)n;
        if (n > 0) { int len = (int)bits_ue(b) + 1; if (len > 32) H_FAIL(d, "offset_len_minus1 out of range"); for (unsigned i = 0; i < n; i++) bits_u(b, len);
            }
    }
    if (pps->sh_extension) { unsigned n = bits_ue(b); if (n > 256) H_FAIL(d, "slice header extension too long"); for (unsigned i = 0; i < n; i++) bits_u(b, 8);
        }
    /* byte_alignment() */
    if (!bits_u1(b)) H_FAIL(d, "slice header: alignment bit missing");
    while (b->pos & 7) bits_u1(b);
    if (b->err) H_FAIL(d, "slice header truncated");
    sh->data_offset = b->pos >> 3;
    return 0;
}
