// jmcodec_amd/csrc/h264_syntax.cpp -- see h264_syntax.h.
#include "h264_syntax.h"

namespace jmamd {

static const uint8_t kDef4Intra[16] = {6,13,13,20,20,20,28,28,28,28,32,32,32,37,37,42};
static const uint8_t kDef4Inter[16] = {10,14,14,20,20,20,24,24,24,24,27,27,27,30,30,34};
static const uint8_t kDef8Intra[64] = {
  6,10,10,13,11,13,16,16,16,16,18,18,18,18,18,23,23,23,23,23,23,25,25,25,25,25,25,25,27,27,27,27,
  27,27,27,27,29,29,29,29,29,29,29,31,31,31,31,31,31,33,33,33,33,33,36,36,36,36,38,38,38,40,40,42};
static const uint8_t kDef8Inter[64] = {
  9,13,13,15,13,15,17,17,17,17,19,19,19,19,19,21,21,21,21,21,21,22,22,22,22,22,22,22,24,24,24,24,
  24,24,24,24,25,25,25,25,25,25,25,27,27,27,27,27,27,28,28,28,28,28,30,30,30,30,32,32,32,33,33,35};

static bool read_scaling_list(BitReader &br, uint8_t *dst, int n) {   // returns useDefaultScalingMatrixFlag
    int last = 8, next = 8; bool use_default = false;
    for (int j = 0; j < n; j++) {
        if (next) { next = (last + br.se() + 256) & 255; if (j == 0 && next == 0) use_default = true; }
        dst[j] = (uint8_t)(next ? next : last);
        last = dst[j];
    }
    return use_default;
}
static void read_scaling_matrix(BitReader &br, uint8_t s4[6][16], uint8_t s8[2][64], int count,
                                const uint8_t (*fb4)[16], const uint8_t (*fb8)[64]) {
    for (int i = 0; i < count; i++) {
        bool present = br.u1();
        if (i < 6) {
            const uint8_t *dflt = i < 3 ? kDef4Intra : kDef4Inter;
            if (present) { if (read_scaling_list(br, s4[i], 16)) memcpy(s4[i], dflt, 16); }
            else if (i == 0 || i == 3) memcpy(s4[i], fb4 ? fb4[i] : dflt, 16);
            else memcpy(s4[i], s4[i - 1], 16);
        } else {
            int k = i - 6; const uint8_t *dflt = k ? kDef8Inter : kDef8Intra;
            if (k > 1) { if (present) { uint8_t tmp[64]; read_scaling_list(br, tmp, 64); } continue; }
            if (present) { if (read_scaling_list(br, s8[k], 64)) memcpy(s8[k], dflt, 64); }
            else memcpy(s8[k], fb8 ? fb8[k] : dflt, 64);
        }
    }
}

int SeqParams::dpb_frames() const {
    int mbs;
    switch (level_idc) {
    case 9: case 10: mbs = 396; break;
    case 11: mbs = (constraint_flags & 0x10) ? 396 : 900; break;
    case 12: case 13: case 20: mbs = 2376; break;
    case 21: mbs = 4752; break;
    case 22: case 30: mbs = 8100; break;
    case 31: mbs = 18000; break;
    case 32: mbs = 20480; break;
    case 40: case 41: mbs = 32768; break;
    case 42: mbs = 34816; break;
    case 50: mbs = 110400; break;
    default: mbs = 184320; break;
    }
    int n = mbs / (mb_w * mb_h);
    if (n > 16) n = 16;
    if (max_dec_frame_buffering >= 0) n = max_dec_frame_buffering;
    if (n < max_num_ref_frames) n = max_num_ref_frames;
    if (n < 1) n = 1;
    if (n > 16) n = 16;
    return n;
}

static void skip_hrd(BitReader &br) {
    int n = (int)br.ue() + 1;
    br.u(4); br.u(4);
    for (int i = 0; i < n && !br.overrun(); i++) { br.ue(); br.ue(); br.u1(); }
    br.u(5); br.u(5); br.u(5); br.u(5);
}

std::string ParamSets::parse_sps(BitReader &br) {
    SeqParams s;
    memset(s.scaling4, 16, sizeof s.scaling4); memset(s.scaling8, 16, sizeof s.scaling8);
    s.profile_idc = br.u(8); s.constraint_flags = br.u(8); s.level_idc = br.u(8);
    { uint32_t id = br.ue(); if (id > 31) return "sps id out of range"; s.id = (int)id; }
    switch (s.profile_idc) {
    case 100: case 110: case 122: case 244: case 44: case 83: case 86: case 118: case 128: case 138: case 139: case 134: case 135:
        { uint32_t cf = br.ue(); if (cf > 3) return "bad chroma_format_idc"; s.chroma_format_idc = (int)cf; }
        if (s.chroma_format_idc == 3) br.u1();
        { uint32_t bl = br.ue(), bc = br.ue(); if (bl > 6 || bc > 6) return "bad bit depth"; s.bit_depth_luma = 8 + (int)bl; s.bit_depth_chroma = 8 + (int)bc; }
        br.u1();
        s.scaling_matrix_present = br.u1();
        if (s.scaling_matrix_present) read_scaling_matrix(br, s.scaling4, s.scaling8, s.chroma_format_idc != 3 ? 8 : 12, nullptr, nullptr);
        break;
    default: break;
    }
    // every ue() below is range-checked as an unsigned value BEFORE it is narrowed: a crafted SPS may carry 0xFFFFFFFE in any field
    // (7.4.2.1.1 ranges: log2_max_frame_num_minus4 / log2_max_pic_order_cnt_lsb_minus4 0..12, max_num_ref_frames 0..16)
    { uint32_t v = br.ue(); if (v > 12) return "log2_max_frame_num out of range"; s.log2_max_frame_num = 4 + (int)v; }
    { uint32_t v = br.ue(); if (v > 2) return "bad pic_order_cnt_type"; s.poc_type = (int)v; }
    if (s.poc_type == 0) { uint32_t v = br.ue(); if (v > 12) return "log2_max_pic_order_cnt_lsb out of range"; s.log2_max_poc_lsb = 4 + (int)v; }
    else if (s.poc_type == 1) {
        s.delta_pic_order_always_zero = br.u1();
        s.offset_for_non_ref_pic = br.se(); s.offset_for_top_to_bottom = br.se();
        { uint32_t v = br.ue(); if (v > 255) return "bad num_ref_frames_in_pic_order_cnt_cycle"; s.num_ref_frames_in_poc_cycle = (int)v; }
        for (int i = 0; i < s.num_ref_frames_in_poc_cycle; i++) s.offset_for_ref_frame[i] = br.se();
    }
    { uint32_t v = br.ue(); if (v > 16) return "max_num_ref_frames out of range"; s.max_num_ref_frames = (int)v; }
    s.gaps_allowed = br.u1();
    { uint32_t w = br.ue(), h = br.ue(); if (w >= 1024 || h >= 1024) return "picture too large"; s.mb_w = (int)w + 1; s.mb_h = (int)h + 1; }
    s.frame_mbs_only = br.u1();
    // frame_mbs_only_flag = 0: the stream MAY hold field pictures (PAFF) or, with mb_adaptive_frame_field_flag, field macroblock pairs (MBAFF).
    // pic_height_in_map_units then counts field macroblock rows: FrameHeightInMbs is twice that (7.4.2.1.1).  Frame pictures of such a stream without
    // MBAFF are coded exactly like progressive ones; field pictures decode as pictures of half the height (decoder.cpp); MBAFF streams are refused here.
    if (!s.frame_mbs_only) { s.mbaff = br.u1(); s.mb_h *= 2; }
    if ((long long)s.mb_w * s.mb_h > 139264) return "picture too large";       // MaxFS of level 6.2 (Table A-1)
    s.direct_8x8_inference = br.u1();
    if (br.u1()) {
        uint32_t c[4]; for (auto &v : c) v = br.ue();
        // frame_crop_*_offset in chroma units (4:2:0 -> 2 luma samples, 7.4.2.1.1): the cropped picture must keep at least one sample
        if (c[0] > 8192 || c[1] > 8192 || c[2] > 8192 || c[3] > 8192 || 2 * (c[0] + c[1]) >= (uint32_t)s.mb_w * 16 ||
            2 * (c[2] + c[3]) >= (uint32_t)s.mb_h * 16) return "frame cropping larger than the picture";
        // (vertical crop units are 2 luma rows of a FRAME, or of a field when frame_mbs_only_flag = 0: CropUnitY = SubHeightC * (2 - frame_mbs_only_flag))
        const int vy = s.frame_mbs_only ? 1 : 2;
        if (2 * vy * (c[2] + c[3]) >= (uint32_t)s.mb_h * 16) return "frame cropping larger than the picture";
        s.crop_l = (int)c[0]; s.crop_r = (int)c[1]; s.crop_t = (int)c[2] * vy; s.crop_b = (int)c[3] * vy;
    }
    if (br.u1()) {   // VUI (E.1.1): the bitstream restriction sizes the DPB; the timing information is what jm_intel_get_stream_info reports
        if (br.u1()) { if (br.u(8) == 255) { br.u(16); br.u(16); } }
        if (br.u1()) br.u1();
        if (br.u1()) { br.u(3); br.u1(); if (br.u1()) { br.u(8); br.u(8); br.u(8); } }
        if (br.u1()) { br.ue(); br.ue(); }
        if (br.u1()) { s.num_units_in_tick = br.u(32); s.time_scale = br.u(32); s.fixed_frame_rate = br.u1(); }   // timing_info_present_flag
        bool nal_hrd = br.u1(); if (nal_hrd) skip_hrd(br);
        bool vcl_hrd = br.u1(); if (vcl_hrd) skip_hrd(br);
        if (nal_hrd || vcl_hrd) br.u1();
        br.u1();
        if (br.u1()) {
            br.u1(); br.ue(); br.ue(); br.ue(); br.ue();
            uint32_t ro = br.ue(), db = br.ue();
            // out-of-range restriction: ignored (falls back to the level's DPB size)
            if (!br.overrun() && ro <= 16 && db <= 16 && ro <= db) { s.max_num_reorder_frames = (int)ro; s.max_dec_frame_buffering = (int)db; }
        }
    }
    if (br.overrun()) return "SPS truncated";
    if (s.chroma_format_idc != 1 || s.bit_depth_luma != 8 || s.bit_depth_chroma != 8) return "only 8-bit 4:2:0 is supported";
    if (!s.frame_mbs_only && s.mbaff) return "interlaced streams with MBAFF are not supported";
    if (!s.frame_mbs_only && !s.direct_8x8_inference) return "direct_8x8_inference_flag must be 1 when frame_mbs_only_flag is 0";
    s.valid = true;
    sps[s.id] = s;
    return "";
}

std::string ParamSets::parse_pps(BitReader &br) {
    PicParamSet p;
    { uint32_t id = br.ue(), sid = br.ue(); if (id > 255 || sid > 31) return "pps/sps id out of range"; p.id = (int)id; p.sps_id = (int)sid; }
    p.cabac = br.u1(); p.bottom_field_poc_present = br.u1();
    if (br.ue() != 0) return "slice groups (FMO) are not supported";
    { uint32_t a = br.ue(), b = br.ue(); if (a > 31 || b > 31) return "num_ref_idx_default_active out of range"; p.num_ref_idx_default[0] = (int)a + 1;
        p.num_ref_idx_default[1] = (int)b + 1; }
    p.weighted_pred = br.u1(); p.weighted_bipred_idc = br.u(2);
    if (p.weighted_bipred_idc > 2) return "bad weighted_bipred_idc";
    { int q = br.se(); if (q < -26 || q > 25) return "pic_init_qp out of range"; p.init_qp = 26 + q; }
    br.se();
    p.chroma_qp_off = br.se();
    if (p.chroma_qp_off < -12 || p.chroma_qp_off > 12) return "chroma_qp_index_offset out of range";
    p.deblock_ctrl_present = br.u1(); p.constrained_intra = br.u1(); p.redundant_pic_cnt_present = br.u1();
    p.second_chroma_qp_off = p.chroma_qp_off;
    const SeqParams *s = sps[p.sps_id].valid ? &sps[p.sps_id] : nullptr;
    if (s) { memcpy(p.scaling4, s->scaling4, sizeof p.scaling4); memcpy(p.scaling8, s->scaling8, sizeof p.scaling8); }
    else { memset(p.scaling4, 16, sizeof p.scaling4); memset(p.scaling8, 16, sizeof p.scaling8); }
    br.set_end_from_trailing();
    if (br.more_rbsp_data()) {
        p.transform8x8 = br.u1();
        p.scaling_matrix_present = br.u1();
        if (p.scaling_matrix_present) {
            bool sps_has = s && s->scaling_matrix_present;
            read_scaling_matrix(br, p.scaling4, p.scaling8, 6 + 2 * (int)p.transform8x8, sps_has ? s->scaling4 : nullptr, sps_has ? s->scaling8 : nullptr);
        }
        p.second_chroma_qp_off = br.se();
        if (p.second_chroma_qp_off < -12 || p.second_chroma_qp_off > 12) return "second_chroma_qp_index_offset out of range";
    }
    if (br.overrun()) return "PPS truncated";
    p.valid = true;
    pps[p.id] = p;
    return "";
}

std::string ParamSets::parse_slice_header(BitReader &br, int nal_type, int nal_ref_idc, SliceHeader &sh) const {
    sh.nal_ref_idc = nal_ref_idc; sh.idr = nal_type == 5;
    { uint32_t v = br.ue(); if (v >= 139264) return "first_mb_in_slice out of range"; sh.first_mb = (int)v; }
    uint32_t st = br.ue();
    if (st > 9) return "bad slice_type";
    sh.type = st % 5;
    if (sh.type > SL_I) return "SP/SI slices are not supported";
    { uint32_t id = br.ue(); if (id > 255 || !pps[id].valid) return "slice refers to a missing PPS"; sh.pps_id = (int)id; }
    const PicParamSet &p = pps[sh.pps_id];
    if (!sps[p.sps_id].valid) return "slice refers to a missing SPS";
    const SeqParams &s = sps[p.sps_id];
    sh.frame_num = br.u(s.log2_max_frame_num);
    if (!s.frame_mbs_only) { sh.field_pic = br.u1(); if (sh.field_pic) sh.bottom_field = br.u1(); }      // field_pic_flag, bottom_field_flag
    if (sh.idr) sh.idr_pic_id = br.ue();
    // the order count of the OTHER field (delta_pic_order_cnt_bottom, delta_pic_order_cnt[1]) is only sent with frames (7.3.3)
    if (s.poc_type == 0) { sh.poc_lsb = br.u(s.log2_max_poc_lsb); if (p.bottom_field_poc_present && !sh.field_pic) sh.delta_poc_bottom = br.se(); }
    else if (s.poc_type == 1 && !s.delta_pic_order_always_zero) { sh.delta_poc[0] = br.se();
        if (p.bottom_field_poc_present && !sh.field_pic) sh.delta_poc[1] = br.se(); }
    if (p.redundant_pic_cnt_present) { const uint32_t r = br.ue(); if (r > 127) return "redundant_pic_cnt out of range"; sh.redundant_pic_cnt = (int)r; }
    if (sh.type == SL_B) sh.direct_spatial_mv_pred = br.u1();
    sh.num_ref_idx[0] = p.num_ref_idx_default[0]; sh.num_ref_idx[1] = p.num_ref_idx_default[1];
    if (sh.type != SL_I && br.u1()) {
        uint32_t a = br.ue(), b = sh.type == SL_B ? br.ue() : 0;
        if (a > (sh.field_pic ? 31u : 15u) || b > (sh.field_pic ? 31u : 15u)) return "num_ref_idx_active out of range";      // 7.4.3
        sh.num_ref_idx[0] = (int)a + 1; if (sh.type == SL_B) sh.num_ref_idx[1] = (int)b + 1;
    }
    if (sh.type != SL_B) sh.num_ref_idx[1] = 0;
    if (sh.type == SL_I) sh.num_ref_idx[0] = 0;
    int lists = sh.type == SL_I ? 0 : (sh.type == SL_B ? 2 : 1);
    for (int l = 0; l < lists; l++) {
        if (!br.u1()) continue;
        for (;;) {
            uint32_t idc = br.ue();
            if (idc == 3) break;
            if (idc > 3 || sh.n_mod[l] >= 66 || br.overrun()) return "bad ref_pic_list_modification";
            sh.mod[l][sh.n_mod[l]++] = RefMod{(uint8_t)idc, br.ue()};
        }
    }
    if ((p.weighted_pred && sh.type == SL_P) || (p.weighted_bipred_idc == 1 && sh.type == SL_B)) {
        sh.explicit_wp = true;
        { uint32_t a = br.ue(), b = br.ue(); if (a > 7 || b > 7) return "bad weight denominator"; sh.luma_log2_wd = (int)a; sh.chroma_log2_wd = (int)b; }
        for (int l = 0; l < lists; l++) for (int i = 0; i < sh.num_ref_idx[l]; i++) {
            int lw = 1 << sh.luma_log2_wd, lo = 0, cw[2] = {1 << sh.chroma_log2_wd, 1 << sh.chroma_log2_wd}, co[2] = {0, 0};
            if (br.u1()) { lw = br.se(); lo = br.se(); }
            if (br.u1()) for (int j = 0; j < 2; j++) { cw[j] = br.se(); co[j] = br.se(); }
            if (lw < -128 || lw > 127 || lo < -128 || lo > 127 || cw[0] < -128 || cw[0] > 127 || cw[1] < -128 || cw[1] > 127 || co[0] < -128 || co[0] > 127 ||
                co[1] < -128 || co[1] > 127) return "weight out of range";
            if (lw != (1 << sh.luma_log2_wd) || lo != 0 || cw[0] != (1 << sh.chroma_log2_wd) || cw[1] != cw[0] || co[0] != 0 ||
                co[1] != 0) sh.wp_nondefault = true;
            sh.luma_w[l][i] = (int16_t)lw; sh.luma_o[l][i] = (int16_t)lo;
            for (int j = 0; j < 2; j++) { sh.chroma_w[l][i][j] = (int16_t)cw[j]; sh.chroma_o[l][i][j] = (int16_t)co[j]; }
        }
    }
    if (nal_ref_idc) {
        if (sh.idr) { br.u1(); sh.long_term_reference = br.u1(); }
        else if ((sh.adaptive_marking = br.u1())) {
            for (;;) {
                uint32_t op = br.ue();
                if (!op) break;
                if (op > 6 || sh.n_mark >= 66 || br.overrun()) return "bad memory_management_control_operation";
                MarkOp m{(uint8_t)op, 0, 0};
                if (op == 1 || op == 3) m.a = br.ue();
                if (op == 2) m.a = br.ue();
                if (op == 3 || op == 6) m.b = br.ue();
                if (op == 4) m.a = br.ue();
                sh.mark[sh.n_mark++] = m;
            }
        }
    }
    if (p.cabac && sh.type != SL_I) { uint32_t v = br.ue(); if (v > 2) return "bad cabac_init_idc"; sh.cabac_init_idc = (int)v; }
    { int d = br.se(); if (d < -51 || d > 51 || p.init_qp + d < 0 || p.init_qp + d > 51) return "slice QP out of range"; sh.qp = p.init_qp + d; }
    if (p.deblock_ctrl_present) {
        { uint32_t v = br.ue(); if (v > 2) return "bad disable_deblocking_filter_idc"; sh.disable_deblock = (int)v; }
        if (sh.disable_deblock != 1) {
            int a = br.se(), b = br.se();
            if (a < -6 || a > 6 || b < -6 || b > 6) return "deblocking filter offset out of range";
            sh.alpha_off = 2 * a; sh.beta_off = 2 * b;
        }
    }
    if (br.overrun()) return "slice header truncated";
    sh.data_bit_offset = br.bitpos();
    return "";
}

}  // namespace jmamd
