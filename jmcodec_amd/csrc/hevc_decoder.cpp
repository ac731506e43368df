// jmcodec_amd/csrc/hevc_decoder.cpp -- the HEVC half of jmamd::Decoder (codec_type 1, /root/reference/nv_dec/nv_dec.h:37-46).
//
// NAL unit layer, picture order count (H.265 8.3.1), reference picture sets (8.3.2), generation of missing references (8.3.3),
// reference picture lists (8.3.4) and output order (C.5.2 "bumping"): what cuvidParseVideoData does for an HEVC stream before it
// calls pfnDecodePicture / pfnDisplayPicture (/root/reference/nv_dec/nv_dec.cpp:23-52, :394).  Splitting, job slots, output slots and the
// hand-over to the device engine are shared with the H.264 path (decoder.cpp).
#include "decoder.h"
#include "engine.h"
#include "kernels.h"
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cstring>

namespace jmamd {

// keep = bytes at the start of the old host buffer that the new one must hold (0: the contents are rebuilt anyway)
bool Decoder::ensure_job_cap(JobSlot &js, size_t bytes, size_t keep) {
    if (bytes <= js.cap) return true;
    if (js.big >= 0) return false;                                     // a borrowed worst-case buffer: nothing bigger exists
    size_t cap = codec_ == 1 ? bytes + bytes / 2 + 4096 : bytes;      // (the H.264 caller chooses its own head room)
    if (parse_only_ || !gpu_open_) { uint8_t *p = (uint8_t *)realloc(js.host, cap); if (!p) return false; js.host = p; js.cap = cap; return true; }
    hipSetDevice(device_);
    uint8_t *h = nullptr, *d = nullptr;
    if (hipHostMalloc((void **)&h, cap, hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&d, cap) != hipSuccess) { if (h) hipHostFree(h);
        return false; }
    if (keep && js.host) memcpy(h, js.host, std::min(keep, js.cap));
    // the slot belongs to the picture being parsed: nothing on the device refers to the old buffers any more (see acquire_job_slot)
    if (js.host) hipHostFree(js.host);
    if (js.dev) hipFree(js.dev);
    js.host = h; js.dev = d; js.cap = cap;
    return true;
}

// ------------------------------------------------------------------------------------------------------------
void Decoder::hevc_handle_nal(const uint8_t *nal, size_t len) {
    if (failed_ || len < 2 || (nal[0] & 0x80)) return;
    const int type = (nal[0] >> 1) & 63, layer = ((nal[0] & 1) << 5) | (nal[1] >> 3), tid = (nal[1] & 7) - 1;
    if (layer != 0 || tid < 0) return;
    const bool is_slice = type <= 9 || (type >= 16 && type <= 21);
    if (!is_slice && type != 33 && type != 34) {
        if (type == 32 || type == 35 || type == 39) hevc_dispatch_pending();            // VPS, AUD, prefix SEI: a new access unit starts
        if (type == 36 || type == 37) {                                              // end of sequence / bitstream: everything decoded so far is output
            hevc_dispatch_pending(); h_seen_eos_ = true;
            if (seq_active_) { for (int i = 0; i < n_surf_; i++) dpb_[i].ref = 0; hevc_bump(carry_out_, true, true);
                for (int i = 0; i < n_surf_; i++) if (!dpb_[i].wait_output) dpb_[i].in_use = false; }
        }
        return;
    }
    std::vector<uint8_t> rbsp(len + Rbsp::kSlack);
    const size_t n = Rbsp::unescape(nal + 2, len - 2, rbsp.data());
    BitReader br(rbsp.data(), n);
    if (!is_slice) {
        hevc_dispatch_pending();
        std::string e = type == 33 ? hps_.parse_sps(br) : hps_.parse_pps(br);
        if (!e.empty()) { stat_errors_++; note_error(e); }
        return;
    }
    const bool first = (rbsp[0] & 0x80) != 0;
    if (first) hevc_dispatch_pending();
    if ((type == 8 || type == 9) && h_no_rasl_output_) return;        // RASL pictures of the CRA / BLA that started decoding are dropped (8.1.3)
    if (h_first_picture_ && !(type >= 16 && type <= 21)) return;      // decoding starts at an IRAP picture
    HevcSliceHeader sh;
    std::string e = hps_.parse_slice_header(br, type, sh, (pending_ && h_have_last_sh_) ? &h_last_sh_ : nullptr);
    if (!e.empty()) { stat_errors_++; note_error(e); return; }
    if (sh.first_in_pic) { if (!hevc_start_picture(sh, type, tid)) return; }
    else if (!pending_ || !pending_->hevc) { stat_errors_++; note_error("slice segment of a picture whose first segment is missing"); return; }
    HevcTask &ht = *pending_->hevc;
    if (!ht.slices.empty() && sh.pps_id != ht.slices[0].sh.pps_id) { stat_errors_++; note_error("slices of one picture refer to different PPSs"); return; }
    if (ht.slices.size() >= 600) { stat_errors_++; return; }
    HevcSliceTask st;
    st.sh = sh;
    if (sh.dependent && !ht.slices.empty()) st.refs = ht.slices.back().refs;
    else if (sh.type != HSL_I) { if (!hevc_build_refs(sh, st.refs)) { stat_errors_++; return; } }
    else { memset(st.refs.slot, -1, sizeof st.refs.slot); memset(st.refs.poc, 0, sizeof st.refs.poc); memset(st.refs.is_lt, 0, sizeof st.refs.is_lt); }
    st.refs.cur_poc = ht.poc;
    st.rbsp = std::move(rbsp); st.len = n;
    ht.slices.push_back(std::move(st));
    h_last_sh_ = sh; h_have_last_sh_ = true;
}

bool Decoder::hevc_activate(const HevcSps &sps) {
    const int mbw = (sps.width + 15) / 16, mbh = (sps.height + 15) / 16;
    const bool changed = !seq_active_ || mbw != mb_w_ || mbh != mb_h_ || sps.width != hsps_.width || sps.height != hsps_.height || sps.disp_w() != disp_w_ ||
        sps.disp_h() != disp_h_;
    h_max_dpb_ = sps.max_dec_pic_buffering; h_reorder_ = sps.max_num_reorder;
    if (!changed) return true;
    if (seq_active_) {
        auto t = std::make_unique<PicTask>();
        t->out_before = std::move(carry_out_); carry_out_.clear();
        push_task(std::move(t));
        { std::unique_lock<std::mutex> lk(mtx_); cv_.wait(lk, [&] { return outstanding_ == 0 && parse_pending_ == 0; }); }
        if (gpu_open_) {
            hipSetDevice(device_);
            free_surfaces();
            if (resid_) { hipFree(resid_); resid_ = nullptr; }
            for (auto &w : hevc_work_) if (w) { hipFree(w); w = nullptr; }
            if (hevc_bs_) { hipFree(hevc_bs_); hevc_bs_ = nullptr; }
            for (auto &j : jobs_) { if (j.host) hipHostFree(j.host); if (j.dev) hipFree(j.dev); if (j.uploaded) hipEventDestroy(j.uploaded); j = JobSlot(); }
            free_out_slots(false);
        } else for (auto &j : jobs_) { free(j.host); j = JobSlot(); }
    }
    mb_w_ = mbw; mb_h_ = mbh; disp_w_ = sps.disp_w(); disp_h_ = sps.disp_h();
    n_surf_ = 18; extra_surf_ = 0;                 // 16 (max DPB) + current + one spare (the pre-SAO work surfaces are separate: hevc_work_)
    for (auto &d : dpb_) d = DpbPic();
    seq_.profile_idc = 100;                        // sizes the shared job buffers (gpu_alloc_sequence)
    if (!gpu_alloc_sequence()) return false;
    seq_active_ = true;
    if (!timer_started_) { t0_ = std::chrono::steady_clock::now(); timer_started_ = true; }
    return true;
}

// C.5.2.2 / C.5.2.3: output pictures in increasing POC until the reorder and size limits hold (all: empty the DPB)
void Decoder::hevc_bump(std::vector<int> &out, bool all, bool use_fullness) {
    for (;;) {
        int n_out = 0, full = 0, best = -1;
        for (int i = 0; i < n_surf_; i++) if (dpb_[i].in_use && i != cur_) { full++; if (dpb_[i].wait_output) { n_out++;
            if (best < 0 || dpb_[i].poc < dpb_[best].poc) best = i; } }
        if (best < 0 || !(all || n_out > h_reorder_ || (use_fullness && full >= h_max_dpb_))) break;
        out.push_back(best); dpb_[best].wait_output = false; display_pocs_.push_back(dpb_[best].poc); dpb_[best].out_at = decode_count_ - 1;
        if (!dpb_[best].ref) dpb_[best].in_use = false;
    }
}

bool Decoder::hevc_start_picture(const HevcSliceHeader &sh, int nal_type, int tid) {
    const HevcPps &pps = hps_.pps[sh.pps_id]; const HevcSps &sps = hps_.sps[pps.sps_id];
    const bool irap = nal_type >= 16 && nal_type <= 23, idr = nal_type == 19 || nal_type == 20;
    const bool no_rasl = irap && (idr || nal_type <= 18 || h_first_picture_ || h_seen_eos_);
    if (irap) h_no_rasl_output_ = no_rasl;
    if (irap && no_rasl && seq_active_) {          // C.5.2.2: a new coded video sequence: everything still waiting is output first (or discarded)
        const bool discard = nal_type == 21 ? true : sh.no_output_of_prior;
        for (int i = 0; i < n_surf_; i++) dpb_[i].ref = 0;
        if (discard) { for (int i = 0; i < n_surf_; i++) { dpb_[i].wait_output = false; dpb_[i].in_use = false; } }
        else hevc_bump(carry_out_, true, true);
        for (int i = 0; i < n_surf_; i++) if (dpb_[i].in_use && !dpb_[i].wait_output) dpb_[i].in_use = false;
    }
    if (!hevc_activate(sps)) return false;
    hsps_ = sps; hpps_ = pps;
    // 8.3.1
    const int max_lsb = 1 << sps.log2_max_poc_lsb;
    int poc = 0;
    if (!idr) {
        int msb = 0;
        if (!(irap && no_rasl)) {
            const int prev_lsb = h_poc_tid0_ & (max_lsb - 1), prev_msb = h_poc_tid0_ - prev_lsb;
            if (sh.poc_lsb < prev_lsb && prev_lsb - sh.poc_lsb >= max_lsb / 2) msb = prev_msb + max_lsb;
            else if (sh.poc_lsb > prev_lsb && sh.poc_lsb - prev_lsb > max_lsb / 2) msb = prev_msb - max_lsb;
            else msb = prev_msb;
        }
        poc = msb + sh.poc_lsb;
    }
    // 8.3.2: every picture that is in none of the five lists stops being a reference
    uint8_t keep[kMaxSurfaces]; memset(keep, 0, sizeof keep);
    if (!idr) {
        for (int i = 0; i < sh.n_lt; i++) for (int k = 0; k < n_surf_; k++) {
            const DpbPic &p = dpb_[k];
            if (!p.in_use || !p.ref || keep[k]) continue;
            if (sh.lt_msb[i] ? p.poc == (poc & ~(max_lsb - 1)) + sh.lt_poc[i] : (p.poc & (max_lsb - 1)) == sh.lt_poc[i]) { keep[k] = 2; break; }
        }
        for (int s = 0; s < 2; s++) for (int i = 0; i < (s ? sh.rps.n_pos : sh.rps.n_neg); i++)
            for (int k = 0; k < n_surf_; k++) if (dpb_[k].in_use && dpb_[k].ref == 1 && !keep[k] && dpb_[k].poc == poc + sh.rps.d[s][i]) { keep[k] = 1; break; }
    }
    for (int k = 0; k < n_surf_; k++) if (dpb_[k].in_use) { dpb_[k].ref = keep[k]; if (!dpb_[k].ref && !dpb_[k].wait_output) dpb_[k].in_use = false; }
    if (!(irap && no_rasl)) hevc_bump(carry_out_, false, true);            // C.5.2.2 "bumping" before the current picture is stored
    // surface for the new picture (cooling rule as for H.264: a surface displayed after picture n is still being packed out during n + 1)
    int slot = -1, warm = -1; bool wait_pack = false;
    for (int i = 0; i < n_surf_; i++) if (!dpb_[i].in_use) { if (decode_count_ >= dpb_[i].out_at + 2) { slot = i; break; } if (warm < 0) warm = i; }
    if (slot < 0 && warm >= 0) { slot = warm; wait_pack = true; }
    if (slot < 0) {                                // non-conformant stream: make room
        int best = -1;
        for (int i = 0; i < n_surf_; i++) if (dpb_[i].wait_output && (best < 0 || dpb_[i].poc < dpb_[best].poc)) best = i;
        if (best >= 0) { carry_out_.push_back(best); display_pocs_.push_back(dpb_[best].poc); dpb_[best].wait_output = false; }
        else { for (int i = 0; i < n_surf_; i++) if (best < 0 || dpb_[i].poc < dpb_[best].poc) best = i; }
        dpb_[best].ref = 0; dpb_[best].in_use = false; slot = best; stat_errors_++;
    }
    cur_ = slot;
    DpbPic &c = dpb_[slot];
    c = DpbPic(); c.in_use = true; c.decode_idx = decode_count_++; c.poc = poc;
    c.hcol = std::make_shared<HevcColMotion>();
    c.wait_output = sh.pic_output && !((nal_type == 8 || nal_type == 9) && h_no_rasl_output_);
    if (tid == 0 && !(nal_type >= 6 && nal_type <= 9) && !(nal_type <= 14 && (nal_type & 1) == 0)) h_poc_tid0_ = poc;
    h_first_picture_ = false; h_seen_eos_ = false; h_have_last_sh_ = false;
    pending_ = std::make_unique<PicTask>();
    pending_->has_picture = true; pending_->cur_slot = slot; pending_->wait_prev_pack = wait_pack;
    pending_->out_before = std::move(carry_out_); carry_out_.clear();
    pending_->hevc = std::make_unique<HevcTask>();
    pending_->hevc->sps = sps; pending_->hevc->pps = pps; pending_->hevc->poc = poc; pending_->hevc->col_out = c.hcol;
    pending_->hevc->work_slot = (int)(hevc_work_rr_++ % kHevcWorkSets);
    first_sh_ = SliceHeader(); first_sh_.type = sh.type == HSL_I ? SL_I : (sh.type == HSL_B ? SL_B : SL_P);
    if (sh.type == HSL_I) stat_i_++; else if (sh.type == HSL_B) stat_b_++; else stat_p_++;
    return true;
}

// 8.3.2 candidate lists + 8.3.3 + 8.3.4 for one slice
bool Decoder::hevc_build_refs(const HevcSliceHeader &sh, HevcSliceRefs &rf) {
    const int max_lsb = 1 << hsps_.log2_max_poc_lsb, poc = dpb_[cur_].poc;
    int before[16], after[16], lt[32], nb = 0, na = 0, nl = 0;
    auto find_st = [&](int want) { for (int k = 0; k < n_surf_; k++) if (k != cur_ && dpb_[k].in_use && dpb_[k].ref == 1 && dpb_[k].poc == want) return k;
        return -1; };
    auto missing = [&](int want, int ref) {       // 8.3.3: a grey stand-in (conformant streams never get here after the first IRAP)
        for (int k = 0; k < n_surf_; k++) if (k != cur_ && !dpb_[k].in_use) {
            dpb_[k] = DpbPic(); dpb_[k].in_use = true; dpb_[k].ref = ref; dpb_[k].poc = want; dpb_[k].decode_idx = -1;
            stat_errors_++;
            return k;
        }
        return -1;
    };
    for (int i = 0; i < sh.rps.n_neg; i++) if (sh.rps.used[0][i]) { int k = find_st(poc + sh.rps.d[0][i]); if (k < 0) k = missing(poc + sh.rps.d[0][i], 1);
        if (k < 0) return false; before[nb++] = k; }
    for (int i = 0; i < sh.rps.n_pos; i++) if (sh.rps.used[1][i]) { int k = find_st(poc + sh.rps.d[1][i]); if (k < 0) k = missing(poc + sh.rps.d[1][i], 1);
        if (k < 0) return false; after[na++] = k; }
    for (int i = 0; i < sh.n_lt; i++) if (sh.lt_used[i]) {
        const int want = sh.lt_msb[i] ? (poc & ~(max_lsb - 1)) + sh.lt_poc[i] : sh.lt_poc[i];
        int k = -1;
        for (int j = 0; j < n_surf_ && k < 0; j++) if (j != cur_ && dpb_[j].in_use && dpb_[j].ref &&
            (sh.lt_msb[i] ? dpb_[j].poc == want : (dpb_[j].poc & (max_lsb - 1)) == want)) k = j;
        if (k < 0) k = missing(want, 2);
        if (k < 0) return false;
        dpb_[k].ref = 2; lt[nl++] = k;
    }
    const int total = nb + na + nl;
    if (total == 0) return false;
    memset(rf.slot, -1, sizeof rf.slot); memset(rf.poc, 0, sizeof rf.poc); memset(rf.is_lt, 0, sizeof rf.is_lt);
    for (int l = 0; l < (sh.type == HSL_B ? 2 : 1); l++) {
        int tmp[64], n = 0; const int want = std::max(sh.n_ref[l], total);
        while (n < want) {
            for (int i = 0; i < (l ? na : nb) && n < want; i++) tmp[n++] = l ? after[i] : before[i];
            for (int i = 0; i < (l ? nb : na) && n < want; i++) tmp[n++] = l ? before[i] : after[i];
            for (int i = 0; i < nl && n < want; i++) tmp[n++] = lt[i];
        }
        for (int i = 0; i < sh.n_ref[l]; i++) { const int k = sh.rplm[l] ? tmp[sh.list_entry[l][i]] : tmp[i]; rf.slot[l][i] = (int8_t)k;
            rf.poc[l][i] = dpb_[k].poc; rf.is_lt[l][i] = dpb_[k].ref == 2; }
    }
    rf.col.reset();
    if (sh.temporal_mvp) { const int k = rf.slot[(sh.type == HSL_B && !sh.col_from_l0) ? 1 : 0][sh.col_ref_idx]; if (k >= 0) rf.col = dpb_[k].hcol; }
    return true;
}

void Decoder::hevc_dispatch_pending() {
    if (!pending_) return;
    std::unique_ptr<PicTask> t = std::move(pending_);
    dpb_[cur_].ref = 1;                            // "used for short-term reference" after decoding (8.3.2 decides later)
    const int done = cur_;
    cur_ = -1;
    hevc_bump(t->out_after, false, false);         // C.5.2.3 additional bumping (the current picture takes part; only the reorder limit applies)
    (void)done;
    t->job_slot = acquire_job_slot();
    push_task(std::move(t));
}

// ------------------------------------------------------------------------------------------------------------
// worker: entropy decode one HEVC picture and pack its job lists
void Decoder::hevc_parse_task(PicTask *t) {
    auto pt0 = std::chrono::steady_clock::now();
    HevcTask &ht = *t->hevc;
    JobSlot &js = jobs_[t->job_slot];
    static thread_local HevcPicParser parser;
    static thread_local HevcPicJobs jobs;
    HevcDigest dg = hdigest_; dg.on = want_digest_;
    // debugging aid: every digest event
    if (dg.on && !dg.trace && getenv("JM_AMD_DEC_DIGEST_TRACE")) dg.trace = fopen(getenv("JM_AMD_DEC_DIGEST_TRACE"), "w");
    parser.begin_picture(ht.sps, ht.pps, ht.poc, &jobs, &dg, ht.col_out.get());
    for (auto &s : ht.slices) {
        std::string e = parser.parse_slice(s.sh, s.refs, s.rbsp.data(), s.len);
        if (!e.empty()) { t->error = e; stat_errors_++; }
        std::vector<uint8_t>().swap(s.rbsp);
    }
    parser.finish_picture();
    if (ht.col_out) ht.col_out->publish();
    if (want_digest_) hdigest_ = dg;
    // pack: every array 16-byte aligned
    size_t off = 0;
    auto place = [&](size_t bytes) { size_t o = off; off = (off + bytes + 15) & ~(size_t)15; return o; };
    ht.off_ctbs = place(jobs.ctbs.size() * sizeof(HevcCtb)); ht.off_qp8 = place(jobs.qp8.size());
    ht.off_pus = place(jobs.pus.size() * sizeof(HevcPu)); ht.off_tbs = place(jobs.tbs.size() * sizeof(HevcTb));
    ht.off_itbs = place(jobs.itbs.size() * sizeof(HevcIntraTb));
    ht.off_coefs = place(jobs.coefs.size() * 4); ht.off_wps = place(jobs.wps.size() * sizeof(HevcWp));
    ht.n_pus = (int)jobs.pus.size(); ht.n_tbs = (int)jobs.tbs.size(); ht.n_itbs = (int)jobs.itbs.size(); ht.any_sao = jobs.any_sao;
    ht.any_deblock = jobs.any_deblock;
    t->n_intra = jobs.n_intra_cu; t->any_deblock = jobs.any_deblock; t->n_slices = (int)ht.slices.size();
    if (!ensure_job_cap(js, off + 64)) { fail("job buffer allocation failed"); }
    else {
        auto put = [&](size_t o, const void *p, size_t bytes) { if (bytes) memcpy(js.host + o, p, bytes); };
        put(ht.off_ctbs, jobs.ctbs.data(), jobs.ctbs.size() * sizeof(HevcCtb)); put(ht.off_qp8, jobs.qp8.data(), jobs.qp8.size());
        put(ht.off_pus, jobs.pus.data(), jobs.pus.size() * sizeof(HevcPu));
        put(ht.off_tbs, jobs.tbs.data(), jobs.tbs.size() * sizeof(HevcTb));
        put(ht.off_itbs, jobs.itbs.data(), jobs.itbs.size() * sizeof(HevcIntraTb)); put(ht.off_coefs, jobs.coefs.data(), jobs.coefs.size() * 4);
        put(ht.off_wps, jobs.wps.data(), jobs.wps.size() * sizeof(HevcWp));
        t->upload_bytes = off;
        if (want_job_digest_) {                              // tests (pictures are parsed in order: sync option): the arrays as the device gets them
            uint64_t h = job_digest_;
            auto eat = [&](size_t o, size_t n) { const uint8_t *b = js.host + o; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
            eat(ht.off_ctbs, jobs.ctbs.size() * sizeof(HevcCtb)); eat(ht.off_qp8, jobs.qp8.size());
            eat(ht.off_pus, jobs.pus.size() * sizeof(HevcPu)); eat(ht.off_tbs, jobs.tbs.size() * sizeof(HevcTb));
            eat(ht.off_itbs, jobs.itbs.size() * sizeof(HevcIntraTb)); eat(ht.off_coefs, jobs.coefs.size() * 4);
            eat(ht.off_wps, jobs.wps.size() * sizeof(HevcWp));
            job_digest_ = h;
        }
        stat_pictures_++; stat_job_bytes_ += (long long)off; stat_intra_mbs_ += jobs.n_intra_cu; stat_coef_ += (long long)jobs.coefs.size();
        if (!parse_only_ && !failed_) t->upload_seq = engine_->upload(js.dev, js.host, off, js.uploaded, false);
    }
    {
        long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - pt0).count();
        (!ht.slices.empty() && ht.slices[0].sh.type == HSL_I ? stat_parse_ns_i_ : stat_parse_ns_p_) += ns;
    }
    t->state.store(1, std::memory_order_release);
    submit_ready();
    { std::lock_guard<std::mutex> lk(mtx_); parse_pending_--; cv_.notify_all(); }
}

// describe the picture to the device engine (called by submit_task)
void Decoder::hevc_fill_engine_pic(PicTask *t, EnginePic &ep) {
    const HevcTask &ht = *t->hevc;
    JobSlot &js = jobs_[t->job_slot];
    HevcPicParams &hp = ep.hp;
    memset(&hp, 0, sizeof hp);
    ep.codec = 1; ep.uploaded = js.uploaded; ep.upload_seq = t->upload_seq;
    hp.w = ht.sps.width; hp.h = ht.sps.height; hp.pitch = pitch_; hp.chroma_offset = chroma_off_;
    hp.ctb_log2 = ht.sps.log2_ctb; hp.ctb_w = (hp.w + (1 << hp.ctb_log2) - 1) >> hp.ctb_log2; hp.ctb_h = (hp.h + (1 << hp.ctb_log2) - 1) >> hp.ctb_log2;
    hp.w8 = hp.w >> 3; hp.cb_qp_off = ht.pps.cb_qp_off; hp.cr_qp_off = ht.pps.cr_qp_off; hp.strong_intra = ht.sps.strong_intra;
    hp.cur = t->cur_slot; hp.work = ht.work_slot;
    hp.work_surf = ht.any_sao ? hevc_work_[ht.work_slot] : surf_[t->cur_slot];
    for (int i = 0; i < kMaxSurfaces; i++) hp.surf[i] = surf_[i];
    hp.ctbs = (const HevcCtb *)(js.dev + ht.off_ctbs); hp.qp8 = js.dev + ht.off_qp8;
    { uint8_t *set = hevc_bs_ + (size_t)ht.work_slot * hevc_bs_set_bytes_;
      hp.pu_map = (uint32_t *)(set + hevc_bs_off_[0]); hp.cell_flags = set + hevc_bs_off_[1]; hp.bs_v = set + hevc_bs_off_[2]; hp.bs_h = set + hevc_bs_off_[3]; }
    hp.pus = (const HevcPu *)(js.dev + ht.off_pus); hp.n_pus = ht.n_pus; hp.tbs = (const HevcTb *)(js.dev + ht.off_tbs); hp.n_tbs = ht.n_tbs;
    hp.itbs = (const HevcIntraTb *)(js.dev + ht.off_itbs); hp.n_itbs = ht.n_itbs; hp.coefs = (const uint32_t *)(js.dev + ht.off_coefs);
    hp.wps = (const HevcWp *)(js.dev + ht.off_wps); hp.resid = (int16_t *)(resid_ + (size_t)ht.work_slot * ((size_t)mb_w_ * mb_h_ * 768));
    hp.stages = (ht.n_pus ? HPS_MC : 0) | (ht.n_tbs ? HPS_RESID : 0) | (ht.n_itbs ? HPS_INTRA : 0) | (ht.any_deblock ? HPS_DEBLOCK : 0) |
        (ht.any_sao ? HPS_SAO : 0);
    // which surfaces the picture reads: lets the engine put INDEPENDENT pictures of this handle (the B pictures of one pyramid level) into one batch
    for (const auto &sl : ht.slices) for (int l = 0; l < 2; l++) for (int i = 0; i < sl.sh.n_ref[l] && i < 16; i++) if (sl.refs.slot[l][i] >= 0 &&
        sl.refs.slot[l][i] < 32) ep.ref_mask |= 1u << sl.refs.slot[l][i];
    const long long S = (long long)surf_bytes_;
    ep.alg_bytes[0] = (ht.n_pus ? 2 * S : 0) + (long long)t->upload_bytes; ep.alg_bytes[1] = ht.n_itbs ? S : 0;
    ep.alg_bytes[2] = (ht.any_deblock ? 2 * S : 0) + (ht.any_sao ? 2 * S : 0);
}

}  // namespace jmamd
