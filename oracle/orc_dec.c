/*
 * oracle/orc_dec.c -- CPU ORACLE (test infrastructure only).
 * NAL layer (7.3.1, 7.4.1.2.4), picture order count (8.2.1), reference list
 * construction (8.2.4), reference marking (8.2.5), DPB output/bumping (C.4.5)
 * and the reference wrapper's display-area crop.
 * Restates: cuvidParseVideoData + the three parser callbacks
 * (/root/reference/nv_dec/nv_dec.cpp:23-52, :394), the EOS flush (:389-392)
 * and nvdec_create_decoder's target rectangle (:513-519).
 */
#include "orc_internal.h"

/* ------------------------------ DPB helpers ------------------------------ */
static int level_max_dpb_mbs(int level_idc, int constraint_flags) {
    switch (level_idc) {
    case 9: case 10: return 396;
    case 11: return (constraint_flags & 0x10) ? 396 : 900;   /* level 1b signalled via constraint_set3 */
    case 12: case 13: case 20: return 2376;
    case 21: return 4752;
    case 22: case 30: return 8100;
    case 31: return 18000;
    case 32: return 20480;
    case 40: case 41: return 32768;
    case 42: return 34816;
    case 50: return 110400;
    default: return 184320;
    }
}

static void free_pictures(OrcDec *d) {
    for (int i = 0; i <= ORC_MAX_DPB; i++) {
        free(d->dpb[i].y); free(d->dpb[i].u); free(d->dpb[i].v); free(d->dpb[i].mbs);
        memset(&d->dpb[i], 0, sizeof d->dpb[i]);
    }
    d->cur = d->cur_store = d->pending = NULL;
}

/* is_ref of a frame store from the marking of its two fields (see Picture) */
void orc_sync_ref(Picture *p) {
    p->is_ref = (p->fmark[0] == 1 && p->fmark[1] == 1) ? 1 : (p->fmark[0] == 2 && p->fmark[1] == 2) ? 2 : (p->fmark[0] || p->fmark[1]) ? 3 : 0;
}
static void set_ref(Picture *p, int v) { p->fmark[0] = p->fmark[1] = v; p->is_ref = v; }      /* both fields of a store at once */
static int any_short(const Picture *p) { return p->fmark[0] == 1 || p->fmark[1] == 1; }
static int any_long(const Picture *p) { return p->fmark[0] == 2 || p->fmark[1] == 2; }

static void emit(OrcDec *d, Picture *p) {
    const Sps *s = d->asps;
    OrcFrame f;
    int cw = d->mb_w * 16, ch = d->mb_h * 16;
    int dw = cw - 2 * (s->crop_left + s->crop_right), dh = ch - 2 * (s->crop_top + s->crop_bottom);
    if (dw <= 0 || dh <= 0) { dw = cw; dh = ch; }
    /* nv_dec.cpp:513-519: target size = display_area size, display_area origin forced to (0,0) */
    f.y = p->y; f.u = p->u; f.v = p->v;
    f.width = dw; f.height = dh; f.stride_y = p->stride_y; f.stride_c = p->stride_c;
    f.poc = p->poc; f.frame_type = p->frame_type; f.decode_index = p->decode_index;
    p->needed_for_output = 0;
    if (d->cb) d->cb(d->user, &f);
}

static Picture *smallest_poc_waiting(OrcDec *d, const Picture *exclude) {
    Picture *best = NULL;
    for (int i = 0; i <= ORC_MAX_DPB; i++) {
        Picture *p = &d->dpb[i];
        if (!p->in_use || p == exclude || !p->needed_for_output || p->waiting_second) continue;
        if (!best || p->poc < best->poc) best = p;
    }
    return best;
}
static void release_unused(OrcDec *d) {
    for (int i = 0; i <= ORC_MAX_DPB; i++) {
        Picture *p = &d->dpb[i];
        if (p->in_use && p != d->cur_store && p != d->pending && !p->is_ref && !p->needed_for_output) p->in_use = 0;
    }
}
void orc_output_all(OrcDec *d) {
    Picture *p;
    while ((p = smallest_poc_waiting(d, d->cur_store)) != NULL) emit(d, p);
    release_unused(d);
}

static int activate(OrcDec *d, const Sps *sps, const Pps *pps) {
    int changed = !d->asps || d->mb_w != sps->mb_width || d->mb_h != sps->mb_height;
    d->asps = sps; d->apps = pps;
    int frame_mbs = sps->mb_width * sps->mb_height;
    int size = level_max_dpb_mbs(sps->level_idc, sps->constraint_flags) / frame_mbs;
    if (size > 16) size = 16;
    if (sps->max_dec_frame_buffering >= 0) size = sps->max_dec_frame_buffering;
    if (size < sps->max_num_ref_frames) size = sps->max_num_ref_frames;
    if (size < 1) size = 1;
    if (size > 16) size = 16;
    d->dpb_size = size;
    if (changed) {
        free_pictures(d);
        d->mb_w = sps->mb_width; d->mb_h = sps->mb_height;
        d->width = d->mb_w * 16; d->height = d->mb_h * 16;
        for (int i = 0; i <= ORC_MAX_DPB; i++) {
            Picture *p = &d->dpb[i];
            p->stride_y = d->width; p->stride_c = d->width / 2;
            p->y = (uint8_t *)malloc((size_t)d->width * d->height);
            p->u = (uint8_t *)malloc((size_t)d->width * d->height / 4);
            p->v = (uint8_t *)malloc((size_t)d->width * d->height / 4);
            p->mbs = (MbInfo *)malloc(sizeof(MbInfo) * frame_mbs);
            if (!p->y || !p->u || !p->v || !p->mbs) ORC_FAIL(d, "out of memory");
        }
    }
    return 0;
}

/* 8.2.1 decoding process for picture order count: TopFieldOrderCnt and / or BottomFieldOrderCnt of the current frame or field into fpoc[] of its
 * store; returns PicOrderCnt of the picture (a frame: Min(top, bottom); a field: its own) */
static int compute_poc(OrcDec *d, const SliceHdr *sh, Picture *store) {
    const Sps *s = d->asps;
    int max_frame_num = 1 << s->log2_max_frame_num;
    int top = 0, bot = 0;
    if (s->poc_type == 0) {
        int max_lsb = 1 << s->log2_max_poc_lsb, prev_msb, prev_lsb;
        /* prevPicOrderCntMsb / Lsb belong to the previous REFERENCE picture in decoding order (non-reference pictures in between change nothing; the
         * first field of a reference frame is that picture for its second field).  When that picture carried operation 5 they are 0 and its
         * TopFieldOrderCnt after the operation: orc_finish_picture stores exactly that. */
        if (sh->idr) { prev_msb = 0; prev_lsb = 0; }
        else { prev_msb = d->prev_poc_msb; prev_lsb = d->prev_poc_lsb; }
        int msb;
        if (sh->poc_lsb < prev_lsb && prev_lsb - sh->poc_lsb >= max_lsb / 2) msb = prev_msb + max_lsb;
        else if (sh->poc_lsb > prev_lsb && sh->poc_lsb - prev_lsb > max_lsb / 2) msb = prev_msb - max_lsb;
        else msb = prev_msb;
        /* 8.2.1.1: a frame or a top field: TopFieldOrderCnt = msb + lsb; a frame's bottom field lies delta_pic_order_cnt_bottom from it, a bottom
         * field picture has BottomFieldOrderCnt = msb + lsb */
        top = bot = msb + sh->poc_lsb;
        if (!sh->field_pic) bot = top + sh->delta_poc_bottom;
        if (sh->nal_ref_idc) { d->prev_poc_msb = msb; d->prev_poc_lsb = sh->poc_lsb; }
    } else {
        int prev_off = d->prev_ref_has_mmco5 ? 0 : d->prev_frame_num_offset;
        int prev_fn = d->prev_ref_has_mmco5 ? 0 : d->prev_frame_num;
        int off = sh->idr ? 0 : (prev_fn > sh->frame_num ? prev_off + max_frame_num : prev_off);
        d->prev_frame_num_offset = off;
        if (s->poc_type == 2) {
            /* 8.2.1.3: both fields of a frame -- coded as a frame or as two fields -- get the same count */
            top = bot = sh->idr ? 0 : (sh->nal_ref_idc ? 2 * (off + sh->frame_num) : 2 * (off + sh->frame_num) - 1);
        } else {
            int abs_fn = s->num_ref_frames_in_poc_cycle ? off + sh->frame_num : 0;
            if (!sh->nal_ref_idc && abs_fn > 0) abs_fn--;
            int expected = 0, cycle_sum = 0;
            for (int i = 0; i < s->num_ref_frames_in_poc_cycle; i++) cycle_sum += s->offset_for_ref_frame[i];
            if (abs_fn > 0) {
                int cnt = (abs_fn - 1) / s->num_ref_frames_in_poc_cycle, in_cycle = (abs_fn - 1) % s->num_ref_frames_in_poc_cycle;
                expected = cnt * cycle_sum;
                for (int i = 0; i <= in_cycle; i++) expected += s->offset_for_ref_frame[i];
            }
            if (!sh->nal_ref_idc) expected += s->offset_for_non_ref_pic;
            /* 8.2.1.2: frame: top = expected + delta[0], bottom = top + offset_for_top_to_bottom_field + delta[1]; a bottom FIELD picture:
             * expected + offset_for_top_to_bottom_field + delta[0] */
            top = expected + sh->delta_poc[0];
            bot = sh->field_pic ? expected + s->offset_for_top_to_bottom + sh->delta_poc[0] : top + s->offset_for_top_to_bottom + sh->delta_poc[1];
        }
    }
    d->cur_top_poc = top; d->cur_bot_poc = bot;
    if (!sh->field_pic) { store->fpoc[0] = top; store->fpoc[1] = bot; return orc_min(top, bot); }
    store->fpoc[sh->bottom_field] = sh->bottom_field ? bot : top;
    return store->fpoc[sh->bottom_field];
}

/* the view of field `par` of a store: the same samples, every second line (see Picture) */
static Picture *field_view(OrcDec *d, Picture *s, int par) {
    Picture *v = &d->fview[s - d->dpb][par];
    int half = d->mb_w * (d->asps->mb_height / 2);
    *v = *s;
    v->y = s->y + par * s->stride_y; v->u = s->u + par * s->stride_c; v->v = s->v + par * s->stride_c;
    v->stride_y = 2 * s->stride_y; v->stride_c = 2 * s->stride_c;
    v->mbs = s->mbs + par * half;
    v->is_field = 1; v->parity = par; v->store = s;
    v->is_ref = s->fmark[par]; v->poc = s->fpoc[par];
    v->id = 0x40000000 + 2 * s->id + par;       /* its own identity: two fields of one frame are different reference pictures (8.7.2.1) */
    return v;
}

static void store_done(OrcDec *d, Picture *cur);

/* a free frame buffer; C.4.5.3: when there is none, the pictures first in output order go until there is (frames inferred from gaps in frame_num take
 * buffers without passing through the output process) */
static Picture *take_free_picture(OrcDec *d) {
    release_unused(d);
    for (;;) {
        for (int i = 0; i <= ORC_MAX_DPB; i++) if (!d->dpb[i].in_use) return &d->dpb[i];
        Picture *w = smallest_poc_waiting(d, NULL);
        if (!w) return NULL;
        emit(d, w); release_unused(d);
    }
}

/* 8.2.5.2: a frame_num that no picture carried stands for a frame that was not sent.  The frame is inferred: the sliding window runs as for any reference
 * frame without marking operations, and the frame stays in the buffer as a short-term reference "non-existing" -- in the lists, never output, and not to
 * be predicted from.  The order-count state of types 1 and 2 follows the frame numbers. */
static int infer_frame(OrcDec *d, int fn) {
    const int max_frame_num = 1 << d->asps->log2_max_frame_num;
    int nst = 0, nlt = 0; Picture *oldest = NULL;
    for (int i = 0; i <= ORC_MAX_DPB; i++) {
        Picture *p = &d->dpb[i];
        if (!p->in_use) continue;
        p->frame_num_wrap = p->frame_num > fn ? p->frame_num - max_frame_num : p->frame_num;
        if (any_short(p)) { nst++; if (!oldest || p->frame_num_wrap < oldest->frame_num_wrap) oldest = p; }
        else if (any_long(p)) nlt++;
    }
    if (nst + nlt >= orc_max(d->asps->max_num_ref_frames, 1) && oldest) set_ref(oldest, 0);
    Picture *f = take_free_picture(d);
    if (!f) ORC_FAIL(d, "DPB overflow (no room for an inferred frame)");
    f->in_use = 1; set_ref(f, 1); f->needed_for_output = 0; f->has_mmco5 = 0; f->have = 3; f->waiting_second = 0; f->non_existing = 1; f->coded_fields = 0;
    f->id = d->next_pic_id++; f->frame_num = fn; f->is_idr = 0; f->long_term_frame_idx = -1; f->poc = f->fpoc[0] = f->fpoc[1] = 0; f->is_field = 0; f->store = f;
    { const int n = d->mb_w * d->asps->mb_height; for (int i = 0; i < n; i++) f->mbs[i].slice_num = -1; }
    if (d->asps->poc_type != 0) {
        /* 8.2.5.2, second reading (round 4): with pic_order_cnt_type 1 / 2 the inferred frame gets its order counts by 8.2.1, as a reference frame whose
         * delta_pic_order_cnt[] are 0 (this also moves prevFrameNumOffset); a later B slice sorts it into its initial lists by them */
        SliceHdr ih; memset(&ih, 0, sizeof ih); ih.nal_ref_idc = 1; ih.frame_num = fn;
        f->poc = compute_poc(d, &ih, f);
    }
    d->prev_frame_num = fn; d->prev_ref_has_mmco5 = 0;
    d->prev_ref_frame_num = fn;
    d->stats[ORC_ST_INFERRED_FRAMES]++;
    return 0;
}

int orc_start_picture(OrcDec *d, const SliceHdr *sh) {
    const Pps *pps = &d->pps[sh->pps_id];
    const Sps *sps = &d->sps[pps->sps_id];
    /* 3.30 / 7.4.1.2.4: this picture is the SECOND field of the frame whose first field came just before it -- opposite parity, the same frame_num,
     * not an IDR picture, and a reference field exactly if the first one is */
    Picture *pend = d->pending;
    int second = pend && sh->field_pic && !sh->idr && pend->waiting_second && pend->frame_num == sh->frame_num &&
                 pend->have == (sh->bottom_field ? 1 : 2) && pend->first_was_ref == (sh->nal_ref_idc != 0) && sps == d->asps;
    if (pend && !second) { pend->waiting_second = 0; d->pending = NULL; d->stats[ORC_ST_LONE_FIELD]++; store_done(d, pend); }   /* a field that stays alone */
    if (sh->idr || !d->asps) {
        /* new coded video sequence: output everything that is waiting (no_output_of_prior_pics
         * would discard instead; the reference's CUVID parser displays them, so do we) */
        if (d->asps) {
            for (int i = 0; i <= ORC_MAX_DPB; i++) set_ref(&d->dpb[i], 0);
            orc_output_all(d);
        }
        if (activate(d, sps, pps) < 0) return -1;
    } else {
        if (sps != d->asps && (sps->mb_width != d->mb_w || sps->mb_height != d->mb_h)) ORC_FAIL(d, "SPS change without IDR");
        d->asps = sps; d->apps = pps;
    }
    if (!second && !sh->idr && d->decode_count > 0) {
        /* 7.4.3: frame_num is PrevRefFrameNum or the one after it; anything else is a gap (gaps_in_frame_num_value_allowed_flag, or lost pictures) */
        const int max_frame_num = 1 << sps->log2_max_frame_num;
        if (sh->frame_num != d->prev_ref_frame_num && sh->frame_num != (d->prev_ref_frame_num + 1) % max_frame_num)
            for (int fn = (d->prev_ref_frame_num + 1) % max_frame_num; fn != sh->frame_num; fn = (fn + 1) % max_frame_num) if (infer_frame(d, fn) < 0) return -1;
    }
    Picture *cur = NULL;
    if (second) cur = pend;
    else {
        cur = take_free_picture(d);
        if (!cur) ORC_FAIL(d, "DPB overflow (no free picture)");
        cur->in_use = 1; set_ref(cur, 0); cur->needed_for_output = 0; cur->has_mmco5 = 0; cur->have = 0; cur->waiting_second = 0;
        cur->id = d->next_pic_id++; cur->decode_index = d->decode_count++;
        cur->frame_num = sh->frame_num; cur->is_idr = sh->idr; cur->long_term_frame_idx = -1;
        cur->frame_type = sh->slice_type == SLICE_I ? 0 : (sh->slice_type == SLICE_P ? 1 : 2);
        cur->first_was_ref = sh->nal_ref_idc != 0; cur->coded_fields = sh->field_pic; cur->non_existing = 0;
        cur->is_field = 0; cur->store = cur;
        int n = d->mb_w * d->mb_h;
        for (int i = 0; i < n; i++) cur->mbs[i].slice_num = -1;
        /* missing macroblocks stay visible as mid-grey rather than stale data */
        memset(cur->y, 128, (size_t)d->width * d->height);
        memset(cur->u, 128, (size_t)d->width * d->height / 4);
        memset(cur->v, 128, (size_t)d->width * d->height / 4);
    }
    int poc = compute_poc(d, sh, cur);
    if (!second) cur->poc = poc;
    d->cur_store = cur; d->cur = cur; d->cur_mb_count = 0; d->slice_num = 0;
    d->field_pic = sh->field_pic; d->cur_parity = sh->bottom_field; d->is_second_field = second;
    if (sh->field_pic) {
        d->stats[ORC_ST_FIELD_PICS]++; d->stats[ORC_ST_SECOND_FIELDS] += second;
        /* from here to orc_finish_picture the decoder works on a picture of half the height */
        d->cur = field_view(d, cur, sh->bottom_field);
        d->mb_h = sps->mb_height / 2; d->height = d->mb_h * 16;
    }
    d->first_sh = *sh;
    return 0;
}

/* 8.2.4.2.5: the fields of an ordered list of frame stores, alternating in parity and beginning with the parity of the current field.  A store whose
 * field of the wanted parity is not marked `mark` is passed over; when one parity has run out the remaining fields of the other follow in order. */
static int alternate_fields(OrcDec *d, Picture **stores, int n, int mark, int par, Picture **out, int cnt, int max) {
    int c[2] = {0, 0}, q = par;
    for (;;) {
        while (c[q] < n && stores[c[q]]->fmark[q] != mark) c[q]++;
        if (c[q] < n) { if (cnt < max) out[cnt++] = &d->fview[stores[c[q]] - d->dpb][q]; c[q]++; }
        else { int o = q ^ 1; while (c[o] < n && stores[c[o]]->fmark[o] != mark) c[o]++; if (c[o] >= n) break; }
        q ^= 1;
    }
    return cnt;
}

/* 8.2.4.1 (field picture numbers) + 8.2.4.2.2 / 8.2.4.2.4 / 8.2.4.2.5 + 8.2.4.3 for the P or B slices of a field */
static int build_field_lists(OrcDec *d, const SliceHdr *sh) {
    const int par = sh->bottom_field, max_frame_num = 1 << d->asps->log2_max_frame_num;
    const int nlists = sh->slice_type == SLICE_B ? 2 : 1;
    Picture *st[ORC_MAX_DPB + 1], *lt[ORC_MAX_DPB + 1]; int nst = 0, nlt = 0;
    for (int i = 0; i <= ORC_MAX_DPB; i++) {
        Picture *s = &d->dpb[i];
        if (!s->in_use) continue;
        /* the store of the current frame takes part with its first field (8.2.4.2.2: "when the current field is the second field of a complementary
         * field pair and the first field is marked as used for short-term reference, the first field is included") */
        s->frame_num_wrap = s->frame_num > sh->frame_num ? s->frame_num - max_frame_num : s->frame_num;
        for (int q = 0; q < 2; q++) {
            Picture *v = field_view(d, s, q);
            if (s == d->cur_store && q == par) { v->is_ref = 0; continue; }
            /* 8.2.4.1: PicNum = 2 * FrameNumWrap + 1 for a field of the parity of the current field, 2 * FrameNumWrap for the other;
             * LongTermPicNum likewise from LongTermFrameIdx */
            v->pic_num = 2 * s->frame_num_wrap + (q == par);
            v->long_term_pic_num = 2 * s->long_term_frame_idx + (q == par);
        }
        int m0 = s->fmark[0], m1 = s->fmark[1];
        if (s == d->cur_store) { if (par) m1 = 0; else m0 = 0; }
        if (m0 == 1 || m1 == 1) st[nst++] = s;
        if (m0 == 2 || m1 == 2) lt[nlt++] = s;
    }
    for (int i = 0; i < nst; i++) for (int j = i + 1; j < nst; j++) if (st[j]->frame_num_wrap > st[i]->frame_num_wrap) { Picture *t = st[i]; st[i] = st[j];
        st[j] = t; }
    for (int i = 0; i < nlt; i++) for (int j = i + 1; j < nlt; j++) if (lt[j]->long_term_frame_idx < lt[i]->long_term_frame_idx) { Picture *t = lt[i];
        lt[i] = lt[j]; lt[j] = t; }
    Picture *lists[2][35]; memset(lists, 0, sizeof lists);
    int ninit[2] = {0, 0};
    if (nlists == 1) {
        /* (the current field's own store offers its FIRST field only: the mark of the field being decoded is still 0) */
        ninit[0] = alternate_fields(d, st, nst, 1, par, lists[0], 0, 33);
    } else {
        /* 8.2.4.2.4: the short-term stores by PicOrderCnt around the count of the current FIELD: list 0 takes those not above it in descending order,
         * then the others ascending; list 1 the other way round.  PicOrderCnt of a store: of the frame / complementary field pair (Min of its fields)
         * or of its only field. */
        Picture *ord[2][ORC_MAX_DPB + 1], *before[ORC_MAX_DPB + 1], *after[ORC_MAX_DPB + 1]; int nb = 0, na = 0;
        const int cur_poc = d->cur_store->fpoc[par];
        /* (pic_order_cnt_type 0: frames inferred from a gap in frame_num carry no order count and stay out, as in 8.2.4.2.3) */
        for (int i = 0; i < nst; i++) { if (st[i]->non_existing && d->asps->poc_type == 0) continue;
            if (st[i]->poc <= cur_poc) before[nb++] = st[i]; else after[na++] = st[i]; }
        for (int i = 0; i < nb; i++) for (int j = i + 1; j < nb; j++) if (before[j]->poc > before[i]->poc) { Picture *t = before[i]; before[i] = before[j];
            before[j] = t; }
        for (int i = 0; i < na; i++) for (int j = i + 1; j < na; j++) if (after[j]->poc < after[i]->poc) { Picture *t = after[i]; after[i] = after[j];
            after[j] = t; }
        for (int i = 0; i < nb; i++) { ord[0][i] = before[i]; ord[1][na + i] = before[i]; }
        for (int i = 0; i < na; i++) { ord[0][nb + i] = after[i]; ord[1][i] = after[i]; }
        for (int l = 0; l < 2; l++) ninit[l] = alternate_fields(d, ord[l], nst, 1, par, lists[l], 0, 33);
    }
    for (int l = 0; l < nlists; l++) ninit[l] = alternate_fields(d, lt, nlt, 2, par, lists[l], ninit[l], 33);
    if (nlists == 2 && ninit[1] > 1 && ninit[0] == ninit[1]) {
        int same = 1;
        for (int i = 0; i < ninit[0]; i++) if (lists[0][i] != lists[1][i]) same = 0;
        if (same) { Picture *t = lists[1][0]; lists[1][0] = lists[1][1]; lists[1][1] = t; }
    }
    for (int l = 0; l < nlists; l++) {
    Picture **list = lists[l];
    const int nact = sh->num_ref_idx[l];
    for (int i = nact; i < 35; i++) list[i] = NULL;
    if (d->slice_num == 0) { for (int i = 0; i < nact && list[i]; i++) if (list[i]->is_ref == 2) { d->stats[ORC_ST_FIELD_LONG]++; break; }
        d->stats[ORC_ST_FIELD_RPLM] += sh->rplm_flag[l]; }
    if (sh->rplm_flag[l]) {
        const int cur_pic_num = 2 * sh->frame_num + 1, max_pic_num = 2 * max_frame_num;
        int pred = cur_pic_num, idx = 0;
        for (int k = 0; k < sh->n_rplm[l]; k++) {
            const RplmOp *op = &sh->rplm[l][k];
            Picture *target = NULL;
            if (op->idc < 2) {
                int nowrap;
                if (op->idc == 0) { nowrap = pred - (op->val + 1); if (nowrap < 0) nowrap += max_pic_num; }
                else { nowrap = pred + (op->val + 1); if (nowrap >= max_pic_num) nowrap -= max_pic_num; }
                pred = nowrap;
                int pic_num = nowrap > cur_pic_num ? nowrap - max_pic_num : nowrap;
                for (int i = 0; i < nst; i++) for (int q = 0; q < 2; q++) { Picture *v = &d->fview[st[i] - d->dpb][q];
                    if (v->is_ref == 1 && v->pic_num == pic_num) target = v; }
            } else {
                for (int i = 0; i < nlt; i++) for (int q = 0; q < 2; q++) { Picture *v = &d->fview[lt[i] - d->dpb][q];
                    if (v->is_ref == 2 && v->long_term_pic_num == op->val) target = v; }
            }
            if (!target) ORC_FAIL(d, "ref_pic_list_modification names a missing field");
            if (idx >= nact) ORC_FAIL(d, "too many ref_pic_list_modification operations");
            for (int c = nact; c > idx; c--) list[c] = list[c - 1];
            list[idx++] = target;
            int nidx = idx;
            for (int c = idx; c <= nact; c++) if (list[c] != target) list[nidx++] = list[c];
        }
    }
    for (int i = 0; i < nact && i < 33; i++) d->ref_list[l][i] = list[i];
    d->ref_count[l] = nact;
    }
    if (nlists == 2 && d->ref_list[1][0] && !d->asps->direct_8x8_inference) ORC_FAIL(d, "field pictures need direct_8x8_inference_flag");
    if (nlists == 2) d->stats[ORC_ST_B_FIELDS] += d->slice_num == 0;
    return 0;
}

/* 8.2.4.1 + 8.2.4.2.1 + 8.2.4.3: RefPicList0 for P slices of frames */
int orc_build_ref_lists(OrcDec *d, const SliceHdr *sh) {
    d->ref_count[0] = d->ref_count[1] = 0;
    memset(d->ref_list, 0, sizeof d->ref_list);
    if (sh->slice_type == SLICE_I) return 0;
    if (sh->field_pic) return build_field_lists(d, sh);
    int max_frame_num = 1 << d->asps->log2_max_frame_num;
    Picture *st[ORC_MAX_DPB + 1], *lt[ORC_MAX_DPB + 1]; int nst = 0, nlt = 0;
    for (int i = 0; i <= ORC_MAX_DPB; i++) {
        Picture *p = &d->dpb[i];
        if (!p->in_use || p == d->cur) continue;
        if (p->is_ref == 3 && d->slice_num == 0) d->stats[ORC_ST_HALF_STORE]++;
        if (p->is_ref == 1) {
            p->frame_num_wrap = p->frame_num > sh->frame_num ? p->frame_num - max_frame_num : p->frame_num;
            p->pic_num = p->frame_num_wrap; st[nst++] = p;
        } else if (p->is_ref == 2) { p->long_term_pic_num = p->long_term_frame_idx; lt[nlt++] = p; }
    }
    for (int i = 0; i < nst; i++) for (int j = i + 1; j < nst; j++) if (st[j]->pic_num > st[i]->pic_num) { Picture *t = st[i]; st[i] = st[j]; st[j] = t; }
    for (int i = 0; i < nlt; i++) for (int j = i + 1; j < nlt; j++) if (lt[j]->long_term_pic_num < lt[i]->long_term_pic_num) { Picture *t = lt[i];
        lt[i] = lt[j]; lt[j] = t; }
    int nlists = sh->slice_type == SLICE_B ? 2 : 1;
    Picture *init[2][34]; int ninit[2] = {0, 0};
    memset(init, 0, sizeof init);
    if (sh->slice_type == SLICE_P) {                       /* 8.2.4.2.1: descending PicNum, then ascending LongTermPicNum */
        for (int i = 0; i < nst && ninit[0] < 33; i++) init[0][ninit[0]++] = st[i];
    } else {                                               /* 8.2.4.2.3: by POC distance around the current picture */
        Picture *before[ORC_MAX_DPB + 1], *after[ORC_MAX_DPB + 1]; int nb = 0, na = 0;
        /* (a frame inferred from a gap in frame_num has no order count when pic_order_cnt_type is 0: it is not in the initial lists of a B slice) */
        for (int i = 0; i < nst; i++) { if (st[i]->non_existing && d->asps->poc_type == 0) continue;
            if (st[i]->poc < d->cur->poc) before[nb++] = st[i]; else after[na++] = st[i]; }
        for (int i = 0; i < nb; i++) for (int j = i + 1; j < nb; j++) if (before[j]->poc > before[i]->poc) { Picture *t = before[i]; before[i] = before[j];
            before[j] = t; }
        for (int i = 0; i < na; i++) for (int j = i + 1; j < na; j++) if (after[j]->poc < after[i]->poc) { Picture *t = after[i]; after[i] = after[j];
            after[j] = t; }
        for (int i = 0; i < nb; i++) init[0][ninit[0]++] = before[i];
        for (int i = 0; i < na; i++) init[0][ninit[0]++] = after[i];
        for (int i = 0; i < na; i++) init[1][ninit[1]++] = after[i];
        for (int i = 0; i < nb; i++) init[1][ninit[1]++] = before[i];
    }
    for (int l = 0; l < nlists; l++) for (int i = 0; i < nlt && ninit[l] < 33; i++) init[l][ninit[l]++] = lt[i];
    if (nlists == 2 && ninit[1] > 1 && ninit[0] == ninit[1]) {
        int same = 1;
        for (int i = 0; i < ninit[0]; i++) if (init[0][i] != init[1][i]) same = 0;
        if (same) { Picture *t = init[1][0]; init[1][0] = init[1][1]; init[1][1] = t; }
    }
    if (nlists == 2 && init[1][0] == NULL) { /* nothing to check */ }
    for (int l = 0; l < nlists; l++) {
        Picture **list = init[l];
        int nact = sh->num_ref_idx[l];
        for (int i = nact; i < 34; i++) list[i] = NULL;
        if (sh->rplm_flag[l]) {
            int pred = sh->frame_num, idx = 0;
            for (int k = 0; k < sh->n_rplm[l]; k++) {
                const RplmOp *op = &sh->rplm[l][k];
                Picture *target = NULL;
                if (op->idc < 2) {
                    int nowrap;
                    if (op->idc == 0) { nowrap = pred - (op->val + 1); if (nowrap < 0) nowrap += max_frame_num; }
                    else { nowrap = pred + (op->val + 1); if (nowrap >= max_frame_num) nowrap -= max_frame_num; }
                    pred = nowrap;
                    int pic_num = nowrap > sh->frame_num ? nowrap - max_frame_num : nowrap;
                    for (int i = 0; i < nst; i++) if (st[i]->pic_num == pic_num) target = st[i];
                } else {
                    for (int i = 0; i < nlt; i++) if (lt[i]->long_term_pic_num == op->val) target = lt[i];
                }
                if (!target) ORC_FAIL(d, "ref_pic_list_modification names a missing picture");
                if (idx >= nact) ORC_FAIL(d, "too many ref_pic_list_modification operations");
                for (int c = nact; c > idx; c--) list[c] = list[c - 1];
                list[idx++] = target;
                int nidx = idx;
                for (int c = idx; c <= nact; c++) if (list[c] != target) list[nidx++] = list[c];
            }
        }
        for (int i = 0; i < nact; i++) d->ref_list[l][i] = list[i];
        d->ref_count[l] = nact;
    }
    /* 8.4.1.2.1 Fld_To_Frm needs both fields of the colocated pair */
    if (nlists == 2 && d->ref_list[1][0] && d->ref_list[1][0]->coded_fields && d->ref_list[1][0]->have != 3) ORC_FAIL(d, "colocated field pair incomplete");
    return 0;
}

/* 8.2.5 decoded reference picture marking of a FRAME picture */
static int mark_current(OrcDec *d) {
    Picture *cur = d->cur_store; const SliceHdr *sh = &d->first_sh;
    int max_frame_num = 1 << d->asps->log2_max_frame_num;
    if (!sh->nal_ref_idc) return 0;
    if (sh->idr) {
        for (int i = 0; i <= ORC_MAX_DPB; i++) if (&d->dpb[i] != cur) set_ref(&d->dpb[i], 0);
        if (sh->long_term_reference_flag) { set_ref(cur, 2); cur->long_term_frame_idx = 0; d->max_long_term_frame_idx = 0; }
        else { set_ref(cur, 1); d->max_long_term_frame_idx = -1; }
        return 0;
    }
    /* refresh PicNum of short-term pictures relative to the current frame_num */
    for (int i = 0; i <= ORC_MAX_DPB; i++) {
        Picture *p = &d->dpb[i];
        if (p->in_use && p != cur && any_short(p)) {
            p->frame_num_wrap = p->frame_num > sh->frame_num ? p->frame_num - max_frame_num : p->frame_num;
            p->pic_num = p->frame_num_wrap;
        }
    }
    int made_long = 0;
    if (sh->adaptive_marking) {
        /* a frame picture names FRAMES: stores with both fields marked the same way (is_ref 1 / 2; 8.2.4.1) */
        for (int k = 0; k < sh->n_mmco; k++) {
            const Mmco *m = &sh->mmco[k];
            int pic_num_x = sh->frame_num - (m->diff_pic_nums_minus1 + 1);
            for (int i = 0; i <= ORC_MAX_DPB; i++) {
                Picture *p = &d->dpb[i];
                if (!p->in_use || p == cur) continue;
                switch (m->op) {
                case 1: if (p->is_ref == 1 && p->pic_num == pic_num_x) set_ref(p, 0); break;
                case 2: if (p->is_ref == 2 && p->long_term_frame_idx == m->long_term_pic_num) set_ref(p, 0); break;
                case 3:
                    if (p->is_ref == 2 && p->long_term_frame_idx == m->long_term_frame_idx && !(p->pic_num == pic_num_x && 0)) set_ref(p, 0);
                    break;
                case 4: if (any_long(p) && p->long_term_frame_idx > m->max_long_term_frame_idx_plus1 - 1) set_ref(p, 0); break;
                case 5: set_ref(p, 0); break;
                case 6: if (p->is_ref == 2 && p->long_term_frame_idx == m->long_term_frame_idx) set_ref(p, 0); break;
                }
            }
            if (m->op == 3)
                for (int i = 0; i <= ORC_MAX_DPB; i++) {
                    Picture *p = &d->dpb[i];
                    if (p->in_use && p != cur && p->is_ref == 1 && p->pic_num == pic_num_x) { set_ref(p, 2); p->long_term_frame_idx = m->long_term_frame_idx; }
                }
            if (m->op == 4) d->max_long_term_frame_idx = m->max_long_term_frame_idx_plus1 - 1;
            if (m->op == 5) { d->max_long_term_frame_idx = -1; cur->has_mmco5 = 1; }
            if (m->op == 6) { set_ref(cur, 2); cur->long_term_frame_idx = m->long_term_frame_idx; made_long = 1; }
        }
    } else {
        /* 8.2.5.3: numShortTerm + numLongTerm count frames, complementary field pairs and single fields in which ANY field is so marked */
        int nst = 0, nlt = 0; Picture *oldest = NULL;
        for (int i = 0; i <= ORC_MAX_DPB; i++) {
            Picture *p = &d->dpb[i];
            if (!p->in_use || p == cur) continue;
            if (any_short(p)) { nst++; if (!oldest || p->frame_num_wrap < oldest->frame_num_wrap) oldest = p; }
            else if (any_long(p)) nlt++;
        }
        if (nst + nlt >= orc_max(d->asps->max_num_ref_frames, 1) && oldest) set_ref(oldest, 0);
    }
    if (!made_long) set_ref(cur, 1);
    return 0;
}

/* 8.2.5 for a FIELD picture: field picture numbers (8.2.4.1); the sliding window leaves the second field of a reference frame alone (8.2.5.3) */
static int mark_current_field(OrcDec *d) {
    Picture *cur = d->cur_store; const SliceHdr *sh = &d->first_sh;
    const int par = sh->bottom_field, max_frame_num = 1 << d->asps->log2_max_frame_num;
    if (!sh->nal_ref_idc) return 0;
    if (sh->idr) {
        for (int i = 0; i <= ORC_MAX_DPB; i++) if (&d->dpb[i] != cur) set_ref(&d->dpb[i], 0);
        cur->fmark[par] = sh->long_term_reference_flag ? 2 : 1; cur->fmark[par ^ 1] = 0;
        cur->long_term_frame_idx = sh->long_term_reference_flag ? 0 : -1; d->max_long_term_frame_idx = sh->long_term_reference_flag ? 0 : -1;
        orc_sync_ref(cur);
        return 0;
    }
    for (int i = 0; i <= ORC_MAX_DPB; i++) {
        Picture *p = &d->dpb[i];
        if (p->in_use) p->frame_num_wrap = p->frame_num > sh->frame_num ? p->frame_num - max_frame_num : p->frame_num;
    }
    int made_long = 0;
    if (sh->adaptive_marking) {
        const int cur_pic_num = 2 * sh->frame_num + 1;
        for (int k = 0; k < sh->n_mmco; k++) {
            const Mmco *m = &sh->mmco[k];
            const int pic_num_x = cur_pic_num - (m->diff_pic_nums_minus1 + 1);
            if (m->op == 5) ORC_FAIL(d, "memory management operation 5 in a field picture unsupported");
            if (m->op == 3 || m->op == 6) {
                /* 8.2.5.4.3 / 8.2.5.4.6: a field becomes long-term with LongTermFrameIdx idx.  The index is first taken away from whoever holds it --
                 * except from the other field of the SAME frame (the two fields of a long-term pair share their index) */
                Picture *owner = m->op == 6 ? cur : NULL; int owner_q = m->op == 6 ? par : -1;
                if (m->op == 3) for (int i = 0; i <= ORC_MAX_DPB; i++) { Picture *p = &d->dpb[i]; if (!p->in_use) continue;
                    for (int q = 0; q < 2; q++) if (!(p == cur && q == par) && p->fmark[q] == 1 && 2 * p->frame_num_wrap + (q == par) == pic_num_x) {
                        owner = p; owner_q = q; } }
                if (!owner) ORC_FAIL(d, "operation 3 names a missing field");
                for (int i = 0; i <= ORC_MAX_DPB; i++) { Picture *p = &d->dpb[i];
                    if (p->in_use && p != owner && p->long_term_frame_idx == m->long_term_frame_idx) { for (int q = 0; q < 2; q++) if (p->fmark[q] == 2) p->fmark[q] = 0;
                        orc_sync_ref(p); } }
                owner->long_term_frame_idx = m->long_term_frame_idx;
                if (m->op == 3) { owner->fmark[owner_q] = 2; orc_sync_ref(owner); } else made_long = 1;
                d->stats[ORC_ST_FIELD_LONG_OPS]++;
                continue;
            }
            for (int i = 0; i <= ORC_MAX_DPB; i++) {
                Picture *p = &d->dpb[i];
                if (!p->in_use) continue;
                for (int q = 0; q < 2; q++) {
                    if (p == cur && q == par) continue;
                    const int pic_num = 2 * p->frame_num_wrap + (q == par), lt_pic_num = 2 * p->long_term_frame_idx + (q == par);
                    if (m->op == 1 && p->fmark[q] == 1 && pic_num == pic_num_x) { p->fmark[q] = 0; d->stats[ORC_ST_FIELD_MMCO]++; }
                    if (m->op == 2 && p->fmark[q] == 2 && lt_pic_num == m->long_term_pic_num) p->fmark[q] = 0;
                    if (m->op == 4 && p->fmark[q] == 2 && p->long_term_frame_idx > m->max_long_term_frame_idx_plus1 - 1) p->fmark[q] = 0;
                }
                orc_sync_ref(p);
            }
            if (m->op == 4) d->max_long_term_frame_idx = m->max_long_term_frame_idx_plus1 - 1;
        }
    } else if (!(d->is_second_field && cur->fmark[par ^ 1] == 1)) {
        int nst = 0, nlt = 0; Picture *oldest = NULL;
        for (int i = 0; i <= ORC_MAX_DPB; i++) {
            Picture *p = &d->dpb[i];
            if (!p->in_use || p == cur) continue;
            if (any_short(p)) { nst++; if (!oldest || p->frame_num_wrap < oldest->frame_num_wrap) oldest = p; }
            else if (any_long(p)) nlt++;
        }
        if (nst + nlt >= orc_max(d->asps->max_num_ref_frames, 1) && oldest) { set_ref(oldest, 0); d->stats[ORC_ST_FIELD_WINDOW]++; }
    }
    cur->fmark[par] = made_long ? 2 : 1;
    orc_sync_ref(cur);
    return 0;
}

/* C.4.5.2 / C.4.5.3 for a frame store that is complete: a frame, both fields of a frame, or a field whose partner did not come */
static void store_done(OrcDec *d, Picture *cur) {
    if (cur->have != 3) {
        /* a single field: the lines of the missing parity repeat the decoded ones */
        const int par = cur->have == 2, h = d->asps->mb_height * 16;
        for (int y = par; y < h; y += 2) memcpy(cur->y + (size_t)(y ^ 1) * cur->stride_y, cur->y + (size_t)y * cur->stride_y, (size_t)d->width);
        for (int y = par; y < h / 2; y += 2) { memcpy(cur->u + (size_t)(y ^ 1) * cur->stride_c, cur->u + (size_t)y * cur->stride_c, (size_t)d->width / 2);
            memcpy(cur->v + (size_t)(y ^ 1) * cur->stride_c, cur->v + (size_t)y * cur->stride_c, (size_t)d->width / 2); }
    }
    Picture *w = smallest_poc_waiting(d, cur);
    if (!cur->is_ref && (!w || w->poc > cur->poc)) { cur->needed_for_output = 1; emit(d, cur); cur->in_use = 0; release_unused(d); return; }
    cur->needed_for_output = 1;
    for (;;) {
        int used = 0;
        for (int i = 0; i <= ORC_MAX_DPB; i++) { Picture *p = &d->dpb[i]; if (p->in_use && (p->is_ref || p->needed_for_output)) used++; }
        if (used <= d->dpb_size) break;
        Picture *p = smallest_poc_waiting(d, NULL);
        if (!p) break;                       /* DPB full of references: non-conformant, keep going */
        emit(d, p);
    }
    release_unused(d);
}

void orc_finish_picture(OrcDec *d) {
    Picture *cur = d->cur_store;
    if (!cur) return;
    orc_deblock_picture(d, d->cur);
    if (d->field_pic) {
        if (mark_current_field(d) < 0) d->sticky_fail = 1;
        const int par = d->cur_parity;
        d->mb_h = d->asps->mb_height; d->height = d->mb_h * 16;
        d->prev_frame_num = cur->frame_num; d->prev_ref_has_mmco5 = 0;
        if (d->first_sh.nal_ref_idc) d->prev_ref_frame_num = cur->frame_num;
        cur->have |= 1 << par;
        d->cur = d->cur_store = NULL; d->field_pic = 0;
        if (!d->is_second_field) {
            /* the first field of a frame: the store waits for the other one (orc_start_picture decides whether what comes next is that) */
            cur->waiting_second = 1; cur->needed_for_output = 1; d->pending = cur;
            return;
        }
        cur->waiting_second = 0; d->pending = NULL;
        cur->poc = orc_min(cur->fpoc[0], cur->fpoc[1]);                 /* 8.2.1: PicOrderCnt of a complementary field pair */
        store_done(d, cur);
        return;
    }
    cur->have = 3;
    mark_current(d);
    d->prev_frame_num = cur->frame_num;
    d->prev_ref_has_mmco5 = 0;
    if (d->first_sh.nal_ref_idc) d->prev_ref_frame_num = cur->has_mmco5 ? 0 : cur->frame_num;
    if (cur->has_mmco5) {
        /* 7.4.3 / 8.2.1: after operation 5 the picture is inferred to have had frame_num 0, and tempPicOrderCnt = Min(top, bottom) is subtracted
         * from both of its order counts: PicOrderCnt becomes 0, and for pic_order_cnt_type 0 the NEXT pictures see prevPicOrderCntMsb = 0 and
         * prevPicOrderCntLsb = its TopFieldOrderCnt after the subtraction (> 0 when the bottom field lies below the top field).  prev_ref_has_mmco5
         * is the "previous picture" condition of types 1 and 2 (8.2.1.2 / 8.2.1.3: only the picture that follows immediately). */
        d->prev_ref_has_mmco5 = 1; cur->frame_num = 0;
        const int tmp = orc_min(cur->fpoc[0], cur->fpoc[1]);
        if (d->asps->poc_type == 0) { d->prev_poc_msb = 0; d->prev_poc_lsb = cur->fpoc[0] - tmp; }
        Picture *p; while ((p = smallest_poc_waiting(d, cur)) != NULL) emit(d, p);
        cur->fpoc[0] -= tmp; cur->fpoc[1] -= tmp;
        cur->poc = 0;
    }
    d->cur = d->cur_store = NULL;
    store_done(d, cur);
}

/* ------------------------------- NAL layer ------------------------------- */
static int same_picture(const SliceHdr *a, const SliceHdr *b) {       /* 7.4.1.2.4 */
    if (a->frame_num != b->frame_num || a->pps_id != b->pps_id) return 0;
    if (a->field_pic != b->field_pic || a->bottom_field != b->bottom_field) return 0;
    if ((a->nal_ref_idc == 0) != (b->nal_ref_idc == 0)) return 0;
    if (a->poc_lsb != b->poc_lsb || a->delta_poc_bottom != b->delta_poc_bottom) return 0;
    if (a->delta_poc[0] != b->delta_poc[0] || a->delta_poc[1] != b->delta_poc[1]) return 0;
    if (a->idr != b->idr) return 0;
    if (a->idr && a->idr_pic_id != b->idr_pic_id) return 0;
    return 1;
}

OrcDec *orc_open(orc_frame_cb cb, void *user) {
    OrcDec *d = (OrcDec *)calloc(1, sizeof(OrcDec));
    if (!d) return NULL;
    d->cb = cb; d->user = user; d->max_long_term_frame_idx = -1;
    return d;
}
void orc_close(OrcDec *d) {
    if (!d) return;
    free_pictures(d); free(d->rbsp); free(d);
}
const char *orc_last_error(const OrcDec *d) { return d->err; }
/* which coding tools the decoded stream exercised (macroblock / slice counts) */
const char *orc_tool_name(int i) {
    static const char *nm[ORC_ST_N] = {"I4x4", "I8x8", "I16x16", "I_PCM", "P_Skip", "P16x16", "P16x8", "P8x16", "P8x8", "sub<8x8", "T8x8-inter",
        "cabac-slices", "cavlc-slices", "idc0", "idc1", "idc2", "ref>0", "B_Skip", "B_Direct", "B-inter", "exact-slice-ends",
        "field-pictures", "second-fields", "cross-parity-blocks", "field-mmco", "field-rplm", "field-sliding-window", "field-long-term",
        "half-marked-stores", "field-bS3", "field-mvy-limit", "lone-fields", "b-field-pictures", "direct-frame-field-mixed", "field-long-term-ops", "inferred-frames", "redundant-slices-dropped"};
    return i >= 0 && i < ORC_ST_N ? nm[i] : NULL;
}
long orc_tool_count(const OrcDec *d, int i) { return i >= 0 && i < ORC_ST_N ? d->stats[i] : 0; }
void orc_digest_enable(OrcDec *d) { d->digest_on = 1; d->digest = 1469598103934665603ull; d->digest_mbs = 0; }
uint64_t orc_digest_value(const OrcDec *d, uint64_t *n) { if (n) *n = d->digest_mbs; return d->digest; }

int orc_stream_info(const OrcDec *d, int *dw, int *dh, int *cw, int *ch) {
    if (!d->asps) return -1;
    const Sps *s = d->asps;
    int w = d->mb_w * 16, h = d->mb_h * 16;
    if (cw) *cw = w;
    if (ch) *ch = h;
    int x = w - 2 * (s->crop_left + s->crop_right), y = h - 2 * (s->crop_top + s->crop_bottom);
    if (x <= 0 || y <= 0) { x = w; y = h; }
    if (dw) *dw = x;
    if (dh) *dh = y;
    return 0;
}

int orc_decode_nal(OrcDec *d, const uint8_t *nal, size_t len) {
    if (len < 1) return 0;
    if (d->sticky_fail) return -1;
    d->err[0] = 0;
    if (nal[0] & 0x80) ORC_FAIL(d, "forbidden_zero_bit set");
    int ref_idc = (nal[0] >> 5) & 3, type = nal[0] & 31;
    if (len > d->rbsp_cap) { free(d->rbsp); d->rbsp = (uint8_t *)malloc(len + 16); d->rbsp_cap = len; if (!d->rbsp) ORC_FAIL(d, "out of memory"); }
    size_t n = 0; int zeros = 0;
    for (size_t i = 1; i < len; i++) {                /* 7.4.1: drop emulation_prevention_three_byte */
        if (zeros >= 2 && nal[i] == 3) { zeros = 0; continue; }
        d->rbsp[n++] = nal[i];
        zeros = nal[i] == 0 ? zeros + 1 : 0;
    }
    Bits b; bits_init(&b, d->rbsp, n);
    switch (type) {
    case 7: return orc_parse_sps(d, &b);
    case 8: return orc_parse_pps(d, &b);
    case 1: case 5: {
        SliceHdr sh;
        if (orc_parse_slice_header(d, &b, type, ref_idc, &sh) < 0) return -1;
        if (sh.redundant_pic_cnt > 0) { d->stats[ORC_ST_REDUNDANT]++; return 0; }      /* 7.4.3: redundant coded pictures are not decoded */
        if (d->cur && !same_picture(&d->first_sh, &sh)) orc_finish_picture(d);
        if (!d->cur) { if (orc_start_picture(d, &sh) < 0) return -1; }
        else d->slice_num++;
        d->sh = sh;
        if (sh.first_mb >= d->mb_w * d->mb_h) ORC_FAIL(d, "first_mb_in_slice out of range");      /* (mb_h: of the field, in a field picture) */
        if (orc_build_ref_lists(d, &sh) < 0) return -1;
        return orc_decode_slice_data(d, &b);
    }
    case 10: case 11:
        if (d->cur) orc_finish_picture(d);
        return 0;
    default: return 0;                                 /* SEI, AUD, filler, ... */
    }
}

int orc_decode_annexb(OrcDec *d, const uint8_t *buf, size_t len) {
    size_t i = 0; int count = 0;
    /* locate first start code */
    while (i + 3 <= len && !(buf[i] == 0 && buf[i + 1] == 0 && buf[i + 2] == 1)) i++;
    if (i + 3 > len) return 0;
    while (i + 3 <= len) {
        size_t start = i + 3, j = start;
        while (j + 3 <= len && !(buf[j] == 0 && buf[j + 1] == 0 && buf[j + 2] == 1)) j++;
        size_t end = (j + 3 <= len) ? j : len;
        size_t e = end;
        while (e > start && buf[e - 1] == 0) e--;      /* trailing_zero_8bits / 4-byte start code zero */
        if (e > start) { if (orc_decode_nal(d, buf + start, e - start) < 0) return -1; count++; }
        if (j + 3 > len) break;
        i = j;
    }
    return count;
}

void orc_flush(OrcDec *d) {
    if (d->cur) orc_finish_picture(d);
    if (d->pending) { Picture *p = d->pending; p->waiting_second = 0; d->pending = NULL; store_done(d, p); }
    for (int i = 0; i <= ORC_MAX_DPB; i++) set_ref(&d->dpb[i], 0);
    if (d->asps) orc_output_all(d);
}

/* ------------------------- buffer convenience API ------------------------ */
typedef struct { uint8_t *buf; size_t len, cap; int n, w, h, fmt, oom; } Sink;
static void sink_cb(void *user, const OrcFrame *f) {
    Sink *s = (Sink *)user;
    size_t fs = (size_t)f->width * f->height * 3 / 2;
    if (s->len + fs > s->cap) {
        size_t nc = s->cap ? s->cap * 2 : fs * 8;
        while (nc < s->len + fs) nc *= 2;
        uint8_t *nb = (uint8_t *)realloc(s->buf, nc);
        if (!nb) { s->oom = 1; return; }
        s->buf = nb; s->cap = nc;
    }
    uint8_t *o = s->buf + s->len;
    for (int y = 0; y < f->height; y++) memcpy(o + (size_t)y * f->width, f->y + (size_t)y * f->stride_y, f->width);
    o += (size_t)f->width * f->height;
    int cw = f->width / 2, ch = f->height / 2;
    if (s->fmt == 1) {
        for (int y = 0; y < ch; y++) memcpy(o + (size_t)y * cw, f->u + (size_t)y * f->stride_c, cw);
        o += (size_t)cw * ch;
        for (int y = 0; y < ch; y++) memcpy(o + (size_t)y * cw, f->v + (size_t)y * f->stride_c, cw);
    } else {
        for (int y = 0; y < ch; y++) for (int x = 0; x < cw; x++) {
            o[(size_t)y * f->width + 2 * x] = f->u[(size_t)y * f->stride_c + x];
            o[(size_t)y * f->width + 2 * x + 1] = f->v[(size_t)y * f->stride_c + x];
        }
    }
    s->len += fs; s->n++; s->w = f->width; s->h = f->height;
}
int orc_decode_stream_to_buffer(const uint8_t *buf, size_t len, int out_fmt, uint8_t **out, size_t *out_len, int *w, int *h) {
    Sink s; memset(&s, 0, sizeof s); s.fmt = out_fmt;
    OrcDec *d = orc_open(sink_cb, &s);
    if (!d) return -1;
    int rc = orc_decode_annexb(d, buf, len);
    if (rc < 0) fprintf(stderr, "orc: %s\n", orc_last_error(d));
    orc_flush(d);
    if (getenv("ORC_STATS")) {
        fprintf(stderr, "orc tools:");
        for (int i = 0; orc_tool_name(i); i++) if (orc_tool_count(d, i)) fprintf(stderr, " %s=%ld", orc_tool_name(i), orc_tool_count(d, i));
        fprintf(stderr, "\n");
    }
    orc_close(d);
    if (rc < 0 || s.oom) { free(s.buf); return -1; }
    *out = s.buf; *out_len = s.len; if (w) *w = s.w; if (h) *h = s.h;
    return s.n;
}
void orc_free(void *p) { free(p); }
