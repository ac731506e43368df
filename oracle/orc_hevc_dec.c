/*
 * oracle/orc_hevc_dec.c -- CPU ORACLE (test infrastructure only): HEVC NAL layer and picture management.
 * NAL unit header (ITU-T H.265 7.3.1.2), picture order count (8.3.1), reference picture set (8.3.2), generation of
 * unavailable pictures (8.3.3), reference picture lists (8.3.4), output order "bumping" (C.5.2) and the reference
 * wrapper's display-area crop.  Restates cuvidParseVideoData + the three parser callbacks for codec_type 1
 * (/root/reference/nv_dec/nv_dec.cpp:23-52, :394), the EOS flush (:389-392) and the target rectangle (:513-519).
 */
#include "orc_hevc_internal.h"

static const char *kToolNames[HST_N] = { "cu", "intra_cu", "skip_cu", "merge_pu", "amvp_pu", "bi_pu", "amp", "nxn", "tu4", "tu8", "tu16", "tu32", "dst",
    "sign_hiding", "transform_skip", "tq_bypass", "pcm", "cu_qp_delta", "sao_band", "sao_edge", "weighted_pred", "tmvp", "scaling_list", "wpp_rows", "tiles",
    "dependent_slices", "long_term_ref", "rplm", "strong_intra", "constrained_intra", "slices", "i_slices", "p_slices", "b_slices", "merge_b0_b2_vs_pruned_b1",
        "dependent_segment_opens_tile" };
const char *orch_tool_name(int i) { return i >= 0 && i < HST_N ? kToolNames[i] : NULL; }
long orch_tool_count(const OrchDec *d, int i) { return i >= 0 && i < HST_N ? d->stats[i] : 0; }

OrchDec *orch_open(orch_frame_cb cb, void *user) {
    OrchDec *d = calloc(1, sizeof *d);
    if (!d) return NULL;
    d->cb = cb; d->user = user; d->first_picture = 1; d->digest = 0xcbf29ce484222325ULL;
    return d;
}
static void free_pic(HPic *p) { for (int c = 0; c < 3; c++) free(p->pl[c]); free(p->col_mv); free(p->col_ref_poc); free(p->col_ref_lt); free(p->col_intra);
    memset(p, 0, sizeof *p); }
static void free_sequence(OrchDec *d) {
    for (int i = 0; i < H_MAX_DPB; i++) free_pic(&d->dpb[i]);
    free(d->pred_mode); free(d->skip_flag); free(d->ct_depth); free(d->ipm); free(d->nofilter); free(d->qp_y); free(d->edge); free(d->cbf); free(d->mot);
    free(d->slice_of4);
    free(d->ctb_slice_addr); free(d->ctb_slice_idx); free(d->sao); free(d->ctb_rs2ts); free(d->ctb_ts2rs); free(d->tile_id); free(d->col_bd); free(d->row_bd);
    free(d->min_tb_zs);
    for (int c = 0; c < 3; c++) { free(d->deblocked[c]); d->deblocked[c] = NULL; }
    d->pred_mode = d->skip_flag = d->ct_depth = d->ipm = d->nofilter = d->edge = d->cbf = NULL; d->qp_y = NULL; d->mot = NULL; d->slice_of4 = NULL;
    d->ctb_slice_addr = NULL; d->ctb_slice_idx = NULL; d->sao = NULL; d->ctb_rs2ts = d->ctb_ts2rs = d->tile_id = d->col_bd = d->row_bd = NULL;
    d->min_tb_zs = NULL;
    d->cur = NULL;
}
void orch_close(OrchDec *d) { if (!d) return; free_sequence(d); free(d); }
const char *orch_last_error(const OrchDec *d) { return d->err; }
void orch_digest_enable(OrchDec *d) { d->digest_on = 1; }
uint64_t orch_digest_value(const OrchDec *d, uint64_t *n) { if (n) *n = (uint64_t)d->stats[HST_CU]; return d->digest; }
int orch_stream_info(const OrchDec *d, int *dw, int *dh, int *cw, int *ch) {
    if (!d->asps) return -1;
    const HSps *s = d->asps;
    if (dw) *dw = s->width - 2 * (s->conf_win[0] + s->conf_win[1]);
    if (dh) *dh = s->height - 2 * (s->conf_win[2] + s->conf_win[3]);
    if (cw) *cw = s->width;
    if (ch) *ch = s->height;
    return 0;
}

/* ------------------------------ output (C.5.2.4 "bumping") ------------------------------ */
static void emit(OrchDec *d, HPic *p) {
    const HSps *s = d->asps;
    OrchFrame f;
    /* nv_dec.cpp:513-519: target size = display_area size, origin forced to (0,0) */
    f.y = p->pl[0]; f.u = p->pl[1]; f.v = p->pl[2];
    f.width = s->width - 2 * (s->conf_win[0] + s->conf_win[1]); f.height = s->height - 2 * (s->conf_win[2] + s->conf_win[3]);
    f.stride_y = p->stride[0]; f.stride_c = p->stride[1];
    f.poc = p->poc; f.slice_type = p->slice_type; f.decode_index = p->decode_index;
    p->needed_for_output = 0;
    if (d->cb) d->cb(d->user, &f);
}
static int bump(OrchDec *d) {
    HPic *best = NULL;
    for (int i = 0; i < H_MAX_DPB; i++) { HPic *p = &d->dpb[i]; if (p->in_use && p != d->cur && p->needed_for_output && (!best ||
        p->poc < best->poc)) best = p; }
    if (!best) return 0;
    emit(d, best);
    return 1;
}
static void release_unused(OrchDec *d) { for (int i = 0; i < H_MAX_DPB; i++) { HPic *p = &d->dpb[i];
    if (p->in_use && p != d->cur && !p->is_ref && !p->needed_for_output) p->in_use = 0; } }

/* ------------------------------ sequence activation ------------------------------ */
static int activate(OrchDec *d, const HSps *sps, const HPps *pps) {
    int changed = d->asps != sps || d->w != sps->width || d->h != sps->height || d->ctb_size != (1 << sps->log2_ctb);
    if (changed) {
        if (d->asps) { while (bump(d)) {} }
        free_sequence(d);
        d->w = sps->width; d->h = sps->height; d->ctb_size = 1 << sps->log2_ctb;
        d->ctb_w = (d->w + d->ctb_size - 1) >> sps->log2_ctb; d->ctb_h = (d->h + d->ctb_size - 1) >> sps->log2_ctb;
        d->w4 = d->w >> 2; d->h4 = d->h >> 2;
        size_t n4 = (size_t)d->w4 * (size_t)d->h4, nc = (size_t)d->ctb_w * (size_t)d->ctb_h;
        d->pred_mode = calloc(n4, 1); d->skip_flag = calloc(n4, 1); d->ct_depth = calloc(n4, 1); d->ipm = calloc(n4, 1); d->nofilter = calloc(n4, 1);
        d->qp_y = calloc(n4, 1); d->edge = calloc(n4, 1); d->cbf = calloc(n4, 1); d->mot = calloc(n4, sizeof(HMotion));
        d->slice_of4 = calloc(n4, sizeof(int16_t));
        d->ctb_slice_addr = calloc(nc, sizeof(int)); d->ctb_slice_idx = calloc(nc, sizeof(int16_t)); d->sao = calloc(nc, sizeof(HSao));
        d->ctb_rs2ts = calloc(nc, sizeof(int)); d->ctb_ts2rs = calloc(nc, sizeof(int)); d->tile_id = calloc(nc, sizeof(int));
        d->col_bd = calloc(32, sizeof(int)); d->row_bd = calloc(32, sizeof(int));
        d->tb_w = d->ctb_w << (sps->log2_ctb - sps->log2_min_tb); d->tb_h = d->ctb_h << (sps->log2_ctb - sps->log2_min_tb);
        d->min_tb_zs = calloc((size_t)d->tb_w * (size_t)d->tb_h, sizeof(uint32_t));
        for (int c = 0; c < 3; c++) d->deblocked[c] = malloc((size_t)(d->w >> (c ? 1 : 0)) * (size_t)(d->h >> (c ? 1 : 0)));
        if (!d->pred_mode || !d->skip_flag || !d->ct_depth || !d->ipm || !d->nofilter || !d->qp_y || !d->edge || !d->cbf || !d->mot || !d->slice_of4 ||
            !d->ctb_slice_addr ||
            !d->ctb_slice_idx || !d->sao || !d->ctb_rs2ts || !d->ctb_ts2rs || !d->tile_id || !d->col_bd || !d->row_bd || !d->min_tb_zs || !d->deblocked[0] ||
                !d->deblocked[1] || !d->deblocked[2])
            H_FAIL(d, "out of memory");
    }
    d->asps = sps; d->apps = pps;
    /* 6.5.1 CTB raster <-> tile scan, 6.5.2 z-scan order array (depend on the PPS) */
    int nc_ = pps->n_tile_cols, nr_ = pps->n_tile_rows;
    if (nc_ > d->ctb_w || nr_ > d->ctb_h) H_FAIL(d, "more tiles than CTBs");
    int colw[20], rowh[22];
    if (pps->uniform_spacing) { for (int i = 0; i < nc_; i++) colw[i] = ((i + 1) * d->ctb_w) / nc_ - (i * d->ctb_w) / nc_;
        for (int i = 0; i < nr_; i++) rowh[i] = ((i + 1) * d->ctb_h) / nr_ - (i * d->ctb_h) / nr_; }
    else {
        int sum = 0;
        for (int i = 0; i < nc_ - 1; i++) { colw[i] = pps->col_w[i]; sum += colw[i]; } if (sum >= d->ctb_w) H_FAIL(d, "tile columns wider than the picture");
        colw[nc_ - 1] = d->ctb_w - sum;
        sum = 0;
        for (int i = 0; i < nr_ - 1; i++) { rowh[i] = pps->row_h[i]; sum += rowh[i]; } if (sum >= d->ctb_h) H_FAIL(d, "tile rows taller than the picture");
        rowh[nr_ - 1] = d->ctb_h - sum;
    }
    d->col_bd[0] = 0; for (int i = 0; i < nc_; i++) d->col_bd[i + 1] = d->col_bd[i] + colw[i];
    d->row_bd[0] = 0; for (int i = 0; i < nr_; i++) d->row_bd[i + 1] = d->row_bd[i] + rowh[i];
    for (int rs = 0; rs < d->ctb_w * d->ctb_h; rs++) {
        int tx = rs % d->ctb_w, ty = rs / d->ctb_w, tile_x = 0, tile_y = 0;
        for (int i = 0; i < nc_; i++) if (tx >= d->col_bd[i]) tile_x = i;
        for (int i = 0; i < nr_; i++) if (ty >= d->row_bd[i]) tile_y = i;
        int ts = 0;
        for (int i = 0; i < tile_x; i++) ts += rowh[tile_y] * colw[i];
        for (int j = 0; j < tile_y; j++) ts += d->ctb_w * rowh[j];
        ts += (ty - d->row_bd[tile_y]) * colw[tile_x] + tx - d->col_bd[tile_x];
        d->ctb_rs2ts[rs] = ts; d->ctb_ts2rs[ts] = rs; d->tile_id[ts] = tile_y * nc_ + tile_x;
    }
    int sh = sps->log2_ctb - sps->log2_min_tb;
    for (int y = 0; y < d->tb_h; y++) for (int x = 0; x < d->tb_w; x++) {
        int rs = d->ctb_w * (y >> sh) + (x >> sh);
        uint32_t v = (uint32_t)d->ctb_rs2ts[rs] << (sh * 2);
        for (int i = 0; i < sh; i++) { int m = 1 << i; v += (uint32_t)((m & x ? m * m : 0) + (m & y ? 2 * m * m : 0)); }
        d->min_tb_zs[y * d->tb_w + x] = v;
    }
    if (pps->tiles) d->stats[HST_TILES]++;
    return 0;
}

static HPic *alloc_pic(OrchDec *d) {
    for (int i = 0; i < H_MAX_DPB; i++) {
        HPic *p = &d->dpb[i];
        if (p->in_use) continue;
        if (!p->pl[0]) {
            for (int c = 0; c < 3; c++) { p->stride[c] = d->w >> (c ? 1 : 0); p->pl[c] = malloc((size_t)p->stride[c] * (size_t)(d->h >> (c ? 1 : 0))); }
            p->col_w = (d->w + 15) >> 4; p->col_h = (d->h + 15) >> 4;
            size_t n = (size_t)p->col_w * (size_t)p->col_h;
            p->col_mv = calloc(n, sizeof(HMotion)); p->col_ref_poc = calloc(n * 2, sizeof(int)); p->col_ref_lt = calloc(n, 1); p->col_intra = calloc(n, 1);
            if (!p->pl[0] || !p->pl[1] || !p->pl[2] || !p->col_mv || !p->col_ref_poc || !p->col_ref_lt || !p->col_intra) return NULL;
        }
        p->in_use = 1; p->is_ref = 0; p->needed_for_output = 0;
        return p;
    }
    return NULL;
}

/* ------------------------------ picture end ------------------------------ */
static void finish_picture(OrchDec *d) {
    if (!d->pic_started) return;
    d->pic_started = 0;
    HPic *p = d->cur;
    /* macroblocks... coding tree blocks no slice delivered: grey */
    for (int rs = 0; rs < d->ctb_w * d->ctb_h; rs++) if (d->ctb_slice_addr[rs] < 0) {
        int x0 = (rs % d->ctb_w) * d->ctb_size, y0 = (rs / d->ctb_w) * d->ctb_size;
        for (int c = 0; c < 3; c++) { int sc = c ? 1 : 0;
            for (int y = y0 >> sc; y < ((y0 + d->ctb_size) >> sc) && y < (d->h >> sc); y++) for (int x = x0 >> sc; x < ((x0 + d->ctb_size) >> sc) &&
            x < (d->w >> sc); x++) p->pl[c][y * p->stride[c] + x] = 128; }
        for (int y = y0; y < y0 + d->ctb_size && y < d->h; y += 4) for (int x = x0; x < x0 + d->ctb_size && x < d->w; x += 4) {
            int i = (y >> 2) * d->w4 + (x >> 2); d->pred_mode[i] = 2; d->edge[i] = 0; d->nofilter[i] = 1; d->slice_of4[i] = 0; d->qp_y[i] = 26; }
    }
    orch_deblock_picture(d);
    orch_sao_picture(d);
    /* motion storage for temporal prediction (8.5.3.2.8: the 16x16 compressed field) */
    for (int y = 0; y < p->col_h; y++) for (int x = 0; x < p->col_w; x++) {
        int i4 = (y * 4) * d->w4 + x * 4, e = y * p->col_w + x;
        p->col_intra[e] = d->pred_mode[i4] != 1;
        p->col_mv[e] = d->mot[i4];
        const HSlice *sl = &d->slices[d->slice_of4[i4]];
        p->col_ref_lt[e] = 0;
        for (int l = 0; l < 2; l++) if (d->mot[i4].pred_flag >> l & 1) { p->col_ref_poc[e * 2 + l] = sl->ref_poc[l][d->mot[i4].ref_idx[l]];
            p->col_ref_lt[e] |= (uint8_t)(sl->ref_is_lt[l][d->mot[i4].ref_idx[l]] << l); }
    }
    /* C.5.2.3: "additional bumping" */
    if (p->pic_output) p->needed_for_output = 1;
    p->is_ref = 1;
    d->cur = NULL;
    const HSps *s = d->asps;
    int hi = s->max_sub_layers - 1;
    for (;;) {
        int n_out = 0;
        for (int i = 0; i < H_MAX_DPB; i++) if (d->dpb[i].in_use && d->dpb[i].needed_for_output) n_out++;
        if (n_out > s->max_num_reorder[hi]) bump(d); else break;
    }
}

/* ------------------------------ picture start: POC, RPS, bumping ------------------------------ */
static int start_picture(OrchDec *d, const HSlice *sh, int nal_type, int tid) {
    const HSps *sps = &d->sps[d->pps[sh->pps_id].sps_id]; const HPps *pps = &d->pps[sh->pps_id];
    int irap = nal_type >= 16 && nal_type <= 23, idr = nal_type == 19 || nal_type == 20;
    int no_rasl = irap && (idr || nal_type <= 18 || d->first_picture || d->seen_eos);
    if (irap) d->no_rasl_output = no_rasl;
    if (activate(d, sps, pps) < 0) return -1;
    /* 8.3.1 picture order count */
    int max_lsb = 1 << sps->log2_max_poc_lsb, poc;
    if (idr) poc = 0;
    else {
        int msb = 0;
        if (!(irap && no_rasl)) {
            int prev_lsb = d->poc_tid0 & (max_lsb - 1), prev_msb = d->poc_tid0 - prev_lsb;
            if (sh->poc_lsb < prev_lsb && prev_lsb - sh->poc_lsb >= max_lsb / 2) msb = prev_msb + max_lsb;
            else if (sh->poc_lsb > prev_lsb && sh->poc_lsb - prev_lsb > max_lsb / 2) msb = prev_msb - max_lsb;
            else msb = prev_msb;
        }
        poc = msb + sh->poc_lsb;
    }
    /* 8.3.2 reference picture set */
    if (irap && no_rasl) for (int i = 0; i < H_MAX_DPB; i++) d->dpb[i].is_ref = 0;
    uint8_t keep[H_MAX_DPB]; memset(keep, 0, sizeof keep);
    if (!idr) {
        for (int i = 0; i < sh->n_lt; i++) {
            int found = -1;
            for (int k = 0; k < H_MAX_DPB && found < 0; k++) {
                HPic *p = &d->dpb[k];
                if (!p->in_use || !p->is_ref) continue;
                if (sh->lt_msb_present[i] ? p->poc == (poc & ~(max_lsb - 1)) + sh->lt_poc[i] : (p->poc & (max_lsb - 1)) == sh->lt_poc[i]) found = k;
            }
            if (found >= 0) { keep[found] = 2; d->stats[HST_LT_REF]++; }
        }
        for (int s0 = 0; s0 < 2; s0++) for (int i = 0; i < (s0 ? sh->st_rps.n_pos : sh->st_rps.n_neg); i++) {
            int want = poc + sh->st_rps.dpoc[s0][i];
            for (int k = 0; k < H_MAX_DPB; k++) { HPic *p = &d->dpb[k]; if (p->in_use && p->is_ref == 1 && p->poc == want && !keep[k]) { keep[k] = 1; break; } }
        }
    }
    for (int k = 0; k < H_MAX_DPB; k++) if (d->dpb[k].in_use) d->dpb[k].is_ref = keep[k];
    /* C.5.2.2 output and removal of pictures before the current picture is decoded */
    if (irap && no_rasl && !d->first_picture) {
        int no_output = nal_type == 21 ? 1 : sh->no_output_of_prior;
        if (no_output) { for (int i = 0; i < H_MAX_DPB; i++) { d->dpb[i].needed_for_output = 0; } }
        else while (bump(d)) {}
        for (int i = 0; i < H_MAX_DPB; i++) if (d->dpb[i].in_use && !d->dpb[i].is_ref) d->dpb[i].in_use = 0;
    } else {
        release_unused(d);
        int hi = sps->max_sub_layers - 1;
        for (;;) {
            int n_out = 0, full = 0;
            for (int i = 0; i < H_MAX_DPB; i++) if (d->dpb[i].in_use) { full++; if (d->dpb[i].needed_for_output) n_out++; }
            if (n_out > sps->max_num_reorder[hi] || full >= sps->max_dec_pic_buffering[hi]) { if (!bump(d)) break; release_unused(d); } else break;
        }
    }
    HPic *p = alloc_pic(d);
    if (!p) H_FAIL(d, "DPB overflow (reference picture set keeps more pictures than the DPB holds)");
    d->cur = p; d->pic_started = 1;
    p->poc = poc; p->decode_index = d->decode_count++; p->slice_type = sh->type;
    p->pic_output = sh->pic_output && !((nal_type == 8 || nal_type == 9) && d->no_rasl_output);
    if (tid == 0 && !(nal_type >= 6 && nal_type <= 9) && !(nal_type <= 14 && (nal_type & 1) == 0)) d->poc_tid0 = poc;
    d->first_picture = 0; d->seen_eos = 0;
    d->n_slices = 0; d->dep_valid = 0; d->wpp_valid_pic = 0; d->last_cu_qp = 26;
    for (int i = 0; i < d->ctb_w * d->ctb_h; i++) { d->ctb_slice_addr[i] = -1; d->ctb_slice_idx[i] = 0; }
    memset(d->pred_mode, 0, (size_t)d->w4 * (size_t)d->h4);
    memset(d->sao, 0, sizeof(HSao) * (size_t)d->ctb_w * (size_t)d->ctb_h);
    return 0;
}

/* 8.3.2 (candidate lists) + 8.3.3 + 8.3.4 for one slice */
static HPic *find_ref(OrchDec *d, int poc, int lt, int lsb_only, int max_lsb) {
    for (int k = 0; k < H_MAX_DPB; k++) {
        HPic *p = &d->dpb[k];
        if (!p->in_use || p == d->cur || !p->is_ref) continue;
        if (lt ? (lsb_only ? (p->poc & (max_lsb - 1)) == poc : p->poc == poc) : (p->is_ref == 1 && p->poc == poc)) return p;
    }
    return NULL;
}
static HPic *make_missing(OrchDec *d, int poc, int lt) {              /* 8.3.3 generation of unavailable reference pictures */
    HPic *p = alloc_pic(d);
    if (!p) return NULL;
    for (int c = 0; c < 3; c++) memset(p->pl[c], 128, (size_t)p->stride[c] * (size_t)(d->h >> (c ? 1 : 0)));
    memset(p->col_intra, 1, (size_t)p->col_w * (size_t)p->col_h);
    p->poc = poc; p->is_ref = lt ? 2 : 1; p->needed_for_output = 0; p->pic_output = 0; p->decode_index = -1;
    return p;
}
static int build_ref_lists(OrchDec *d, HSlice *sh) {
    const HSps *sps = d->asps;
    int max_lsb = 1 << sps->log2_max_poc_lsb, poc = d->cur->poc;
    HPic *before[16], *after[16], *ltc[32]; int nb = 0, na = 0, nl = 0;
    for (int i = 0; i < sh->st_rps.n_neg; i++) if (sh->st_rps.used[0][i]) { int want = poc + sh->st_rps.dpoc[0][i]; HPic *p = find_ref(d, want, 0, 0, 0);
        if (!p) p = make_missing(d, want, 0); if (!p) H_FAIL(d, "no room for a missing reference picture"); before[nb++] = p; }
    for (int i = 0; i < sh->st_rps.n_pos; i++) if (sh->st_rps.used[1][i]) { int want = poc + sh->st_rps.dpoc[1][i]; HPic *p = find_ref(d, want, 0, 0, 0);
        if (!p) p = make_missing(d, want, 0); if (!p) H_FAIL(d, "no room for a missing reference picture"); after[na++] = p; }
    for (int i = 0; i < sh->n_lt; i++) if (sh->lt_used[i]) {
        int want = sh->lt_msb_present[i] ? (poc & ~(max_lsb - 1)) + sh->lt_poc[i] : sh->lt_poc[i];
        HPic *p = find_ref(d, want, 1, !sh->lt_msb_present[i], max_lsb);
        if (!p) p = make_missing(d, want, 1);
        if (!p) H_FAIL(d, "no room for a missing reference picture");
        p->is_ref = 2; ltc[nl++] = p;
    }
    int total = nb + na + nl;
    if (total == 0) H_FAIL(d, "P/B slice without reference pictures");
    for (int l = 0; l < (sh->type == H_SLICE_B ? 2 : 1); l++) {
        HPic *tmp[64]; int n = 0, want = sh->n_ref[l] > total ? sh->n_ref[l] : total;
        while (n < want) {
            HPic **first = l ? after : before, **second = l ? before : after; int nf = l ? na : nb, ns = l ? nb : na;
            for (int i = 0; i < nf && n < want; i++) tmp[n++] = first[i];
            for (int i = 0; i < ns && n < want; i++) tmp[n++] = second[i];
            for (int i = 0; i < nl && n < want; i++) tmp[n++] = ltc[i];
        }
        for (int i = 0; i < sh->n_ref[l]; i++) {
            HPic *p = sh->rplm_flag[l] ? tmp[sh->list_entry[l][i]] : tmp[i];
            sh->ref_poc[l][i] = p->poc; sh->ref_is_lt[l][i] = p->is_ref == 2; sh->ref_dpb[l][i] = (int8_t)(p - d->dpb);
        }
        if (sh->rplm_flag[l]) d->stats[HST_RPLM]++;
    }
    return 0;
}

/* ------------------------------ NAL layer ------------------------------ */
int orch_decode_nal(OrchDec *d, const uint8_t *nal, size_t len) {
    if (len < 2) return 0;
    int type = (nal[0] >> 1) & 63, layer = ((nal[0] & 1) << 5) | (nal[1] >> 3), tid = (nal[1] & 7) - 1;
    if (layer != 0 || tid < 0) return 0;
    /* 7.3.1.1: remove emulation prevention bytes */
    uint8_t *rbsp = malloc(len + 8); size_t n = 0;
    if (!rbsp) H_FAIL(d, "out of memory");
    for (size_t i = 2; i < len; i++) { if (i >= 4 && nal[i] == 3 && nal[i - 1] == 0 && nal[i - 2] == 0) continue; rbsp[n++] = nal[i]; }
    memset(rbsp + n, 0, 8);
    Bits b; bits_init(&b, rbsp, n);
    int rc = 0;
    if (type == 33) { finish_picture(d); rc = orch_parse_sps(d, &b); }
    else if (type == 34) { finish_picture(d); rc = orch_parse_pps(d, &b); }
    else if (type == 36 || type == 37) {
        /* end of sequence / bitstream: everything decoded so far is output (as at the end of the stream) */
        finish_picture(d); while (bump(d)) {} for (int i = 0; i < H_MAX_DPB; i++) d->dpb[i].is_ref = 0; release_unused(d); d->seen_eos = 1;
    }
    else if (type == 35 || type == 32 || type == 39) { finish_picture(d); }
    else if (type <= 9 || (type >= 16 && type <= 21)) {
        static __thread HSlice sh;
        const HSlice *prev = d->n_slices > 0 && d->pic_started ? &d->slices[d->n_slices - 1] : NULL;
        int first = (int)bits_peek(&b, 1);
        if (first) { finish_picture(d); prev = NULL; }
        if ((type == 8 || type == 9) && d->no_rasl_output) { free(rbsp); return 0; }
            /* RASL pictures of a CRA that starts the stream are not decoded (8.1.3) */
        if (d->first_picture && !(type >= 16 && type <= 21)) { free(rbsp); return 0; }    /* decoding starts at an IRAP picture */
        rc = orch_parse_slice_header(d, &b, type, &sh, prev);
        if (rc == 0 && !sh.first_in_pic && !d->pic_started) { snprintf(d->err, sizeof d->err, "slice segment of a picture whose first segment is missing");
            rc = -1; }
        if (rc == 0 && sh.first_in_pic) rc = start_picture(d, &sh, type, tid);
        if (rc == 0 && d->apps != &d->pps[sh.pps_id]) { snprintf(d->err, sizeof d->err, "slices of one picture refer to different PPSs"); rc = -1; }
        if (rc == 0 && d->n_slices >= H_MAX_SLICES) { snprintf(d->err, sizeof d->err, "too many slice segments"); rc = -1; }
        if (rc == 0) {
            int idx = d->n_slices++;
            d->slices[idx] = sh;
            HSlice *s = &d->slices[idx];
            if (s->dependent) { memcpy(s->ref_poc, prev->ref_poc, sizeof s->ref_poc); memcpy(s->ref_is_lt, prev->ref_is_lt, sizeof s->ref_is_lt);
                memcpy(s->ref_dpb, prev->ref_dpb, sizeof s->ref_dpb); }
            else {
                memset(s->ref_dpb, -1, sizeof s->ref_dpb);
                if (s->type != H_SLICE_I) rc = build_ref_lists(d, s);
                d->stats[HST_SLICES]++; d->stats[HST_I + (s->type == H_SLICE_I ? 0 : s->type == H_SLICE_P ? 1 : 2)]++;
            }
            if (rc == 0) { if (s->type < d->cur->slice_type) d->cur->slice_type = s->type; rc = orch_decode_slice_data(d, s, idx, rbsp, n); }
        }
    }
    free(rbsp);
    return rc;
}

int orch_decode_annexb(OrchDec *d, const uint8_t *buf, size_t len) {
    size_t i = 0, start = (size_t)-1; int count = 0;
    while (i + 3 <= len) {
        if (buf[i] == 0 && buf[i + 1] == 0 && buf[i + 2] == 1) {
            if (start != (size_t)-1) { size_t e = i; while (e > start && buf[e - 1] == 0) e--; if (orch_decode_nal(d, buf + start, e - start) < 0) return -1;
                count++; }
            start = i + 3; i += 3;
        } else i++;
    }
    if (start != (size_t)-1 && start < len) { size_t e = len; while (e > start && buf[e - 1] == 0) e--;
        if (orch_decode_nal(d, buf + start, e - start) < 0) return -1; count++; }
    return count;
}
void orch_flush(OrchDec *d) { finish_picture(d); while (bump(d)) {} release_unused(d); }

/* ------------------------------ convenience: whole stream -> packed frames ------------------------------ */
typedef struct { uint8_t *buf; size_t len, cap; int fmt, n, w, h, oom; } Sink;
static void sink_cb(void *u, const OrchFrame *f) {
    Sink *s = u;
    size_t need = (size_t)f->width * (size_t)f->height * 3 / 2;
    if (s->len + need > s->cap) { size_t nc = s->cap ? s->cap * 2 : need * 4; while (nc < s->len + need) nc *= 2; uint8_t *nb = realloc(s->buf, nc); if (!nb) {
        s->oom = 1; return; } s->buf = nb; s->cap = nc; }
    uint8_t *o = s->buf + s->len;
    for (int y = 0; y < f->height; y++) memcpy(o + (size_t)y * (size_t)f->width, f->y + (size_t)y * (size_t)f->stride_y, (size_t)f->width);
    o += (size_t)f->width * (size_t)f->height;
    int cw = f->width / 2, ch = f->height / 2;
    if (s->fmt == 1) {
        for (int y = 0; y < ch; y++) memcpy(o + (size_t)y * (size_t)cw, f->u + (size_t)y * (size_t)f->stride_c, (size_t)cw);
        o += (size_t)cw * (size_t)ch;
        for (int y = 0; y < ch; y++) memcpy(o + (size_t)y * (size_t)cw, f->v + (size_t)y * (size_t)f->stride_c, (size_t)cw);
    } else for (int y = 0; y < ch; y++) for (int x = 0; x < cw; x++) {
        o[(size_t)y * (size_t)f->width + 2 * (size_t)x] = f->u[(size_t)y * (size_t)f->stride_c + (size_t)x];
        o[(size_t)y * (size_t)f->width + 2 * (size_t)x + 1] = f->v[(size_t)y * (size_t)f->stride_c + (size_t)x]; }
    s->len += need; s->n++; s->w = f->width; s->h = f->height;
}
int orch_decode_stream_to_buffer(const uint8_t *buf, size_t len, int out_fmt, uint8_t **out, size_t *out_len, int *w, int *h) {
    Sink s; memset(&s, 0, sizeof s); s.fmt = out_fmt;
    OrchDec *d = orch_open(sink_cb, &s);
    if (!d) return -1;
    int rc = orch_decode_annexb(d, buf, len);
    if (rc < 0) fprintf(stderr, "orch: %s\n", orch_last_error(d));
    orch_flush(d);
    if (getenv("ORC_STATS")) { fprintf(stderr, "orch tools:"); for (int i = 0; i < HST_N; i++) if (d->stats[i]) fprintf(stderr, " %s=%ld", kToolNames[i],
        d->stats[i]); fprintf(stderr, "\n"); }
    orch_close(d);
    if (rc < 0 || s.oom) { free(s.buf); return -1; }
    *out = s.buf; *out_len = s.len; if (w) *w = s.w; if (h) *h = s.h;
    return s.n;
}
void orch_free(void *p) { free(p); }
