// jmcodec_amd/csrc/hevc_syntax.cpp -- see hevc_syntax.h.
#include "hevc_syntax.h"
#include "hevc_tables.h"

namespace jmamd {

namespace {
int ceil_log2(int v) { int n = 0; while ((1 << n) < v) n++; return n; }

// positions of the up-right diagonal scan (6.5.3) of an n x n block
void diag_positions(int n, uint8_t *px, uint8_t *py) {
    int k = 0;
    for (int s = 0; s <= 2 * (n - 1); s++)
        for (int x = 0; x <= s; x++) { int y = s - x; if (x < n && y < n) { px[k] = (uint8_t)x; py[k] = (uint8_t)y; k++; } }
}

struct CodedLists {                    // scaling lists in coded (diagonal) order; 32x32 uses matrix ids 0 and 3
    uint8_t l[4][6][64]; uint8_t dc[4][6];
    void defaults() {
        for (int m = 0; m < 6; m++) {
            memset(l[0][m], 16, 64);
            for (int s = 1; s < 4; s++) { memcpy(l[s][m], hevc_scaling_default[m >= 3], 64); dc[s][m] = 16; }
        }
    }
    void expand(HevcScaling &sf) const {
        uint8_t x4[16], y4[16], x8[64], y8[64];
        diag_positions(4, x4, y4); diag_positions(8, x8, y8);
        for (int m = 0; m < 6; m++) {
            for (int i = 0; i < 16; i++) sf.f4[m][y4[i] * 4 + x4[i]] = l[0][m][i];
            for (int i = 0; i < 64; i++) {
                sf.f8[m][y8[i] * 8 + x8[i]] = l[1][m][i];
                for (int j = 0; j < 4; j++) sf.f16[m][(y8[i] * 2 + (j >> 1)) * 16 + x8[i] * 2 + (j & 1)] = l[2][m][i];
            }
            sf.f16[m][0] = dc[2][m];
        }
        for (int k = 0; k < 2; k++) {
            for (int i = 0; i < 64; i++) for (int j = 0; j < 16; j++) sf.f32[k][(y8[i] * 4 + (j >> 2)) * 32 + x8[i] * 4 + (j & 3)] = l[3][k * 3][i];
            sf.f32[k][0] = dc[3][k * 3];
        }
    }
};

bool parse_scaling_list_data(BitReader &br, HevcScaling &sf) {          // 7.3.4
    CodedLists c; c.defaults();
    for (int size = 0; size < 4; size++) {
        const int count = size == 3 ? 2 : 6, n = size == 0 ? 16 : 64;
        for (int k = 0; k < count; k++) {
            const int m = size == 3 ? 3 * k : k;
            if (!br.u1()) {
                uint32_t delta = br.ue();
                if (delta > (uint32_t)k) return false;
                if (delta == 0) { if (size == 0) memset(c.l[0][m], 16, 16); else memcpy(c.l[size][m], hevc_scaling_default[size == 3 ? k : (m >= 3)], 64);
                    c.dc[size][m] = 16; }
                else { int rm = size == 3 ? 3 * (k - (int)delta) : m - (int)delta; memmove(c.l[size][m], c.l[size][rm], (size_t)n);
                    c.dc[size][m] = c.dc[size][rm]; }
            } else {
                int next = 8;
                if (size > 1) { int v = br.se(); if (v < -7 || v > 247) return false; next = v + 8; c.dc[size][m] = (uint8_t)next; }
                for (int i = 0; i < n; i++) { int d = br.se(); if (d < -128 || d > 127) return false; next = (next + d + 256) & 255;
                    c.l[size][m][i] = (uint8_t)next; }
            }
        }
    }
    if (br.overrun()) return false;
    c.expand(sf);
    return true;
}

void skip_profile_tier_level(BitReader &br, int sub_layers_minus1) {     // 7.3.3
    br.u(8); br.u(32); br.u(4); br.u(32); br.u(11); br.u1(); br.u(8);
    bool prof[8], lev[8];
    for (int i = 0; i < sub_layers_minus1; i++) { prof[i] = br.u1(); lev[i] = br.u1(); }
    if (sub_layers_minus1 > 0) for (int i = sub_layers_minus1; i < 8; i++) br.u(2);
    for (int i = 0; i < sub_layers_minus1; i++) { if (prof[i]) { br.u(32); br.u(32); br.u(24); } if (lev[i]) br.u(8); }
}

// 7.3.7 + 7.4.8: set number `idx`; sets 0..idx-1 are in `known`
bool parse_rps(BitReader &br, HevcRps &out, int idx, int n_sps, const HevcRps *known) {
    out = HevcRps();
    if (idx != 0 && br.u1()) {
        int back = 1;
        if (idx == n_sps) back = (int)br.ue() + 1;
        if (back > idx) return false;
        const HevcRps &ref = known[idx - back];
        int sign = (int)br.u1(), drps = (int)br.ue() + 1;
        if (sign) drps = -drps;
        const int total = ref.n_neg + ref.n_pos;
        bool used[33], keep[33];
        for (int j = 0; j <= total; j++) { used[j] = br.u1(); keep[j] = used[j] ? true : (bool)br.u1(); }
        // candidate deltas of the reference set in increasing order of POC: positives reversed ... no: walk the spec's three loops
        auto push = [&](int list, int &n, int dp, bool u) { if (n >= 16) return false; out.d[list][n] = (int16_t)dp; out.used[list][n] = u; n++; return true; };
        int n = 0;
        for (int j = ref.n_pos - 1; j >= 0; j--) { int dp = ref.d[1][j] + drps; if (dp < 0 && keep[ref.n_neg + j] && !push(0, n, dp,
            used[ref.n_neg + j])) return false; }
        if (drps < 0 && keep[total] && !push(0, n, drps, used[total])) return false;
        for (int j = 0; j < ref.n_neg; j++) { int dp = ref.d[0][j] + drps; if (dp < 0 && keep[j] && !push(0, n, dp, used[j])) return false; }
        out.n_neg = n; n = 0;
        for (int j = ref.n_neg - 1; j >= 0; j--) { int dp = ref.d[0][j] + drps; if (dp > 0 && keep[j] && !push(1, n, dp, used[j])) return false; }
        if (drps > 0 && keep[total] && !push(1, n, drps, used[total])) return false;
        for (int j = 0; j < ref.n_pos; j++) { int dp = ref.d[1][j] + drps; if (dp > 0 && keep[ref.n_neg + j] && !push(1, n, dp,
            used[ref.n_neg + j])) return false; }
        out.n_pos = n;
    } else {
        uint32_t a = br.ue(), b = br.ue();
        if (a > 16 || b > 16 || a + b > 16) return false;
        out.n_neg = (int)a; out.n_pos = (int)b;
        int acc = 0;
        for (int i = 0; i < out.n_neg; i++) { acc -= (int)br.ue() + 1; if (acc < -32768) return false; out.d[0][i] = (int16_t)acc; out.used[0][i] = br.u1(); }
        acc = 0;
        for (int i = 0; i < out.n_pos; i++) { acc += (int)br.ue() + 1; if (acc > 32767) return false; out.d[1][i] = (int16_t)acc; out.used[1][i] = br.u1(); }
    }
    return !br.overrun();
}
}  // namespace

void HevcScaling::set_default() { CodedLists c; c.defaults(); c.expand(*this); }

std::string HevcParamSets::parse_sps(BitReader &br) {
    HevcSps s;
    br.u(4);
    s.max_sub_layers = (int)br.u(3) + 1;
    br.u1();
    if (s.max_sub_layers > 7) return "SPS: sps_max_sub_layers_minus1 out of range";
    skip_profile_tier_level(br, s.max_sub_layers - 1);
    uint32_t id = br.ue();
    if (id > 15) return "SPS id out of range";
    if (br.ue() != 1) return "unsupported chroma format (4:2:0 only)";
    s.width = (int)br.ue(); s.height = (int)br.ue();
    if (br.u1()) for (int i = 0; i < 4; i++) s.conf[i] = (int)br.ue();
    if (br.ue() != 0 || br.ue() != 0) return "unsupported bit depth (8-bit only)";
    s.log2_max_poc_lsb = (int)br.ue() + 4;
    if (s.log2_max_poc_lsb > 16) return "SPS: log2_max_pic_order_cnt_lsb out of range";
    bool all_layers = br.u1();
    for (int i = all_layers ? 0 : s.max_sub_layers - 1; i < s.max_sub_layers; i++) {       // the values of the highest sub-layer are the ones used
        s.max_dec_pic_buffering = (int)br.ue() + 1; s.max_num_reorder = (int)br.ue(); br.ue();
        if (s.max_dec_pic_buffering > 16 || s.max_num_reorder >= s.max_dec_pic_buffering) return "SPS: DPB parameters out of range";
    }
    s.log2_min_cb = (int)br.ue() + 3; s.log2_ctb = s.log2_min_cb + (int)br.ue();
    s.log2_min_tb = (int)br.ue() + 2; s.log2_max_tb = s.log2_min_tb + (int)br.ue();
    s.depth_inter = (int)br.ue(); s.depth_intra = (int)br.ue();
    if (br.overrun() || s.log2_ctb < 4 || s.log2_ctb > 6 || s.log2_min_cb > s.log2_ctb || s.log2_min_tb >= s.log2_min_cb || s.log2_max_tb > 5 ||
        s.log2_max_tb > s.log2_ctb ||
        s.depth_inter > s.log2_ctb - s.log2_min_tb || s.depth_intra > s.log2_ctb - s.log2_min_tb) return "SPS: block size parameters out of range";
    const int mcb = (1 << s.log2_min_cb) - 1;
    if (s.width <= 0 || s.height <= 0 || s.width > 8192 || s.height > 8192 || (s.width & mcb) ||
        (s.height & mcb)) return "SPS: picture size is not a multiple of the minimum coding block";
    if (2 * (s.conf[0] + s.conf[1]) >= s.width || 2 * (s.conf[2] + s.conf[3]) >= s.height) return "SPS: conformance window larger than the picture";
    s.scaling_enabled = br.u1();
    s.sf.set_default();
    if (s.scaling_enabled) { s.scaling_present = br.u1(); if (s.scaling_present && !parse_scaling_list_data(br, s.sf)) return "SPS: bad scaling_list_data"; }
    s.amp = br.u1(); s.sao = br.u1(); s.pcm = br.u1();
    if (s.pcm) {
        s.pcm_bits_y = (int)br.u(4) + 1; s.pcm_bits_c = (int)br.u(4) + 1;
        s.log2_min_pcm = (int)br.ue() + 3; s.log2_max_pcm = s.log2_min_pcm + (int)br.ue();
        s.pcm_loop_filter_disabled = br.u1();
        if (s.pcm_bits_y > 8 || s.pcm_bits_c > 8 || s.log2_min_pcm < s.log2_min_cb || s.log2_max_pcm > 5 ||
            s.log2_max_pcm > s.log2_ctb) return "SPS: PCM parameters out of range";
    }
    s.n_rps = (int)br.ue();
    if (s.n_rps > 64) return "SPS: too many short-term reference picture sets";
    for (int i = 0; i < s.n_rps; i++) if (!parse_rps(br, s.rps[i], i, s.n_rps, s.rps)) return "SPS: bad st_ref_pic_set";
    s.long_term_present = br.u1();
    if (s.long_term_present) {
        s.n_lt = (int)br.ue();
        if (s.n_lt > 32) return "SPS: too many long-term pictures";
        for (int i = 0; i < s.n_lt; i++) { s.lt_lsb[i] = (uint16_t)br.u(s.log2_max_poc_lsb); s.lt_used[i] = (uint8_t)br.u1(); }
    }
    s.temporal_mvp = br.u1(); s.strong_intra = br.u1();
    if (br.overrun()) return "SPS truncated";
    s.valid = true;                    // VUI and extensions do not influence decoding
    // vui_parameters() (E.2.1) up to vui_timing_info: the picture rate jm_intel_get_stream_info reports (intel_dec.cpp:975-990).  Best effort: a VUI that
    // runs off the end of the NAL unit leaves the rate unknown and the SPS valid.
    if (br.u1()) {
        if (br.u1()) { if (br.u(8) == 255) { br.u(16); br.u(16); } }                             // aspect ratio
        if (br.u1()) br.u1();                                                                   // overscan
        if (br.u1()) { br.u(3); br.u1(); if (br.u1()) { br.u(8); br.u(8); br.u(8); } }          // video signal type
        if (br.u1()) { br.ue(); br.ue(); }                                                      // chroma sample location
        br.u1(); br.u1(); br.u1();                                                              // neutral chroma, field_seq, frame_field_info
        if (br.u1()) { br.ue(); br.ue(); br.ue(); br.ue(); }                                    // default display window
        if (br.u1()) { const uint32_t tick = br.u(32), scale = br.u(32); if (!br.overrun()) { s.num_units_in_tick = tick; s.time_scale = scale; } }
    }
    sps[id] = s;
    return "";
}

std::string HevcParamSets::parse_pps(BitReader &br) {
    HevcPps p;
    uint32_t id = br.ue(), sid = br.ue();
    if (id > 63 || sid > 15) return "PPS id out of range";
    p.sps_id = (int)sid;
    p.dependent_slices = br.u1(); p.output_flag_present = br.u1(); p.n_extra_bits = (int)br.u(3);
    p.sign_hiding = br.u1(); p.cabac_init_present = br.u1();
    p.n_ref_default[0] = (int)br.ue() + 1; p.n_ref_default[1] = (int)br.ue() + 1;
    p.init_qp = 26 + br.se();
    p.constrained_intra = br.u1(); p.transform_skip = br.u1();
    p.cu_qp_delta = br.u1();
    if (p.cu_qp_delta) p.diff_cu_qp_delta_depth = (int)br.ue();
    p.cb_qp_off = br.se(); p.cr_qp_off = br.se();
    p.slice_chroma_qp_offsets = br.u1(); p.weighted_pred = br.u1(); p.weighted_bipred = br.u1(); p.tq_bypass = br.u1();
    p.tiles = br.u1(); p.wpp = br.u1();
    if (p.tiles) {
        p.tile_cols = (int)br.ue() + 1; p.tile_rows = (int)br.ue() + 1;
        if (p.tile_cols > 20 || p.tile_rows > 22) return "PPS: too many tiles";
        p.uniform = br.u1();
        if (!p.uniform) { for (int i = 0; i + 1 < p.tile_cols; i++) p.col_w[i] = (int)br.ue() + 1;
            for (int i = 0; i + 1 < p.tile_rows; i++) p.row_h[i] = (int)br.ue() + 1; }
        p.lf_across_tiles = br.u1();
    }
    p.lf_across_slices = br.u1();
    if (br.u1()) {
        p.deblock_override = br.u1(); p.deblock_disabled = br.u1();
        if (!p.deblock_disabled) { p.beta_off = br.se(); p.tc_off = br.se(); }
    }
    p.scaling_present = br.u1();
    if (p.scaling_present && !parse_scaling_list_data(br, p.sf)) return "PPS: bad scaling_list_data";
    p.lists_modification = br.u1();
    p.log2_par_mrg = (int)br.ue() + 2;
    p.sh_extension = br.u1();
    if (br.overrun()) return "PPS truncated";
    if (p.n_ref_default[0] > 15 || p.n_ref_default[1] > 15 || p.init_qp < 0 || p.init_qp > 51 || p.cb_qp_off < -12 || p.cb_qp_off > 12 || p.cr_qp_off < -12 ||
        p.cr_qp_off > 12 ||
        p.beta_off < -6 || p.beta_off > 6 || p.tc_off < -6 || p.tc_off > 6 || p.diff_cu_qp_delta_depth > 3 ||
            p.log2_par_mrg > 6) return "PPS: parameter out of range";
    p.valid = true;
    pps[id] = p;
    return "";
}

std::string HevcParamSets::parse_slice_header(BitReader &br, int nal_type, HevcSliceHeader &sh, const HevcSliceHeader *prev) const {
    const bool irap = nal_type >= 16 && nal_type <= 23, idr = nal_type == 19 || nal_type == 20;
    bool first = br.u1(), no_output = false;
    if (irap) no_output = br.u1();
    uint32_t pid = br.ue();
    if (pid > 63 || !pps[pid].valid) return "slice refers to a missing PPS";
    const HevcPps &pp = pps[pid];
    if (!sps[pp.sps_id].valid) return "slice refers to a missing SPS";
    const HevcSps &sp = sps[pp.sps_id];
    const int n_ctb = ((sp.width + (1 << sp.log2_ctb) - 1) >> sp.log2_ctb) * ((sp.height + (1 << sp.log2_ctb) - 1) >> sp.log2_ctb);
    bool dependent = false; int addr = 0;
    if (!first) {
        if (pp.dependent_slices) dependent = br.u1();
        addr = (int)br.u(ceil_log2(n_ctb));
        if (addr <= 0 || addr >= n_ctb) return "slice_segment_address out of range";
    }
    if (dependent) { if (!prev) return "dependent slice segment without a preceding slice segment"; sh = *prev; }
    else sh = HevcSliceHeader();
    sh.first_in_pic = first; sh.no_output_of_prior = no_output; sh.pps_id = (int)pid; sh.dependent = dependent; sh.segment_addr = addr;
    if (!dependent) {
        sh.slice_addr = addr;
        br.u(pp.n_extra_bits);
        uint32_t t = br.ue();
        if (t > 2) return "bad slice_type";
        sh.type = (int)t;
        if (irap && sh.type != HSL_I) return "IRAP picture with a non-I slice";
        if (pp.output_flag_present) sh.pic_output = br.u1();
        if (!idr) {
            sh.poc_lsb = (int)br.u(sp.log2_max_poc_lsb);
            if (!br.u1()) { if (!parse_rps(br, sh.rps, sp.n_rps, sp.n_rps, sp.rps)) return "bad st_ref_pic_set in the slice header"; }
            else {
                if (sp.n_rps == 0) return "short_term_ref_pic_set_sps_flag without sets in the SPS";
                int k = sp.n_rps > 1 ? (int)br.u(ceil_log2(sp.n_rps)) : 0;
                if (k >= sp.n_rps) return "short_term_ref_pic_set_idx out of range";
                sh.rps = sp.rps[k];
            }
            if (sp.long_term_present) {
                int from_sps = sp.n_lt > 0 ? (int)br.ue() : 0, own = (int)br.ue();
                if (from_sps > sp.n_lt || from_sps < 0 || own < 0 || from_sps + own > 32) return "too many long-term pictures in the slice header";
                sh.n_lt = from_sps + own;
                int cycle_acc = 0;
                for (int i = 0; i < sh.n_lt; i++) {
                    int lsb;
                    if (i < from_sps) { int k = sp.n_lt > 1 ? (int)br.u(ceil_log2(sp.n_lt)) : 0; if (k >= sp.n_lt) return "lt_idx_sps out of range";
                        lsb = sp.lt_lsb[k]; sh.lt_used[i] = sp.lt_used[k]; }
                    else { lsb = (int)br.u(sp.log2_max_poc_lsb); sh.lt_used[i] = (uint8_t)br.u1(); }
                    sh.lt_msb[i] = (uint8_t)br.u1();
                    if (i == 0 || i == from_sps) cycle_acc = 0;                       // (7-52)
                    if (sh.lt_msb[i]) cycle_acc += (int)br.ue();
                    sh.lt_poc[i] = sh.lt_msb[i] ? lsb - (cycle_acc << sp.log2_max_poc_lsb) : lsb;
                }
            }
            if (sp.temporal_mvp) sh.temporal_mvp = br.u1();
        }
        if (sp.sao) { sh.sao_luma = br.u1(); sh.sao_chroma = br.u1(); }
        if (sh.type != HSL_I) {
            sh.n_ref[0] = pp.n_ref_default[0]; sh.n_ref[1] = sh.type == HSL_B ? pp.n_ref_default[1] : 0;
            if (br.u1()) { sh.n_ref[0] = (int)br.ue() + 1; if (sh.type == HSL_B) sh.n_ref[1] = (int)br.ue() + 1; }
            if (sh.n_ref[0] > 15 || sh.n_ref[1] > 15 || sh.n_ref[0] < 1) return "num_ref_idx_active out of range";
            int total = 0;
            for (int s = 0; s < 2; s++) for (int i = 0; i < (s ? sh.rps.n_pos : sh.rps.n_neg); i++) total += sh.rps.used[s][i];
            for (int i = 0; i < sh.n_lt; i++) total += sh.lt_used[i];
            if (total == 0) return "P/B slice with an empty reference picture set";
            if (pp.lists_modification && total > 1) {
                const int nb = ceil_log2(total);
                for (int l = 0; l < (sh.type == HSL_B ? 2 : 1); l++) {
                    sh.rplm[l] = br.u1();
                    if (sh.rplm[l]) for (int i = 0; i < sh.n_ref[l]; i++) { uint32_t e = br.u(nb); if ((int)e >= total) return "list_entry out of range";
                        sh.list_entry[l][i] = (uint8_t)e; }
                }
            }
            if (sh.type == HSL_B) sh.mvd_l1_zero = br.u1();
            if (pp.cabac_init_present) sh.cabac_init = br.u1();
            if (sh.temporal_mvp) {
                if (sh.type == HSL_B) sh.col_from_l0 = br.u1();
                if (sh.n_ref[sh.col_from_l0 ? 0 : 1] > 1) sh.col_ref_idx = (int)br.ue();
                if (sh.col_ref_idx >= sh.n_ref[sh.col_from_l0 ? 0 : 1]) return "collocated_ref_idx out of range";
            }
            if ((pp.weighted_pred && sh.type == HSL_P) || (pp.weighted_bipred && sh.type == HSL_B)) {         // 7.3.6.3
                sh.wp_denom[0] = (int)br.ue(); sh.wp_denom[1] = sh.wp_denom[0] + br.se();
                if (sh.wp_denom[0] > 7 || sh.wp_denom[1] < 0 || sh.wp_denom[1] > 7) return "bad pred_weight_table";
                for (int l = 0; l < (sh.type == HSL_B ? 2 : 1); l++) {
                    bool fy[16], fc[16];
                    for (int i = 0; i < sh.n_ref[l]; i++) fy[i] = br.u1();
                    for (int i = 0; i < sh.n_ref[l]; i++) fc[i] = br.u1();
                    for (int i = 0; i < sh.n_ref[l]; i++) {
                        for (int c = 0; c < 3; c++) { sh.wp_w[l][i][c] = (int16_t)(1 << sh.wp_denom[c ? 1 : 0]); sh.wp_o[l][i][c] = 0; }
                        if (fy[i]) { int dw = br.se(), o = br.se(); if (dw < -128 || dw > 127 || o < -128 || o > 127) return "bad pred_weight_table";
                            sh.wp_w[l][i][0] = (int16_t)(sh.wp_w[l][i][0] + dw); sh.wp_o[l][i][0] = (int16_t)o; }
                        if (fc[i]) for (int c = 1; c < 3; c++) {
                            int dw = br.se(), dofs = br.se();
                            if (dw < -128 || dw > 127 || dofs < -512 || dofs > 511) return "bad pred_weight_table";
                            int w = (1 << sh.wp_denom[1]) + dw, o = 128 + dofs - ((128 * w) >> sh.wp_denom[1]);
                            sh.wp_w[l][i][c] = (int16_t)w; sh.wp_o[l][i][c] = (int16_t)(o < -128 ? -128 : (o > 127 ? 127 : o));
                        }
                    }
                }
                sh.has_wp = true;
            }
            sh.max_merge = 5 - (int)br.ue();
            if (sh.max_merge < 1 || sh.max_merge > 5) return "five_minus_max_num_merge_cand out of range";
        }
        sh.qp = pp.init_qp + br.se();
        if (pp.slice_chroma_qp_offsets) { sh.cb_qp_off = br.se(); sh.cr_qp_off = br.se(); }
        if (sh.qp < 0 || sh.qp > 51 || sh.cb_qp_off < -12 || sh.cb_qp_off > 12 || sh.cr_qp_off < -12 || sh.cr_qp_off > 12) return "slice QP out of range";
        sh.deblock_disabled = pp.deblock_disabled; sh.beta_off = pp.beta_off; sh.tc_off = pp.tc_off;
        if (pp.deblock_override && br.u1()) {
            sh.deblock_disabled = br.u1();
            if (!sh.deblock_disabled) { sh.beta_off = br.se(); sh.tc_off = br.se();
                if (sh.beta_off < -6 || sh.beta_off > 6 || sh.tc_off < -6 || sh.tc_off > 6) return "slice deblocking offsets out of range"; }
        }
        sh.lf_across_slices = pp.lf_across_slices;
        if (pp.lf_across_slices && (sh.sao_luma || sh.sao_chroma || !sh.deblock_disabled)) sh.lf_across_slices = br.u1();
    }
    if (pp.tiles || pp.wpp) {                   // entry points are not needed: substreams are found by decoding
        uint32_t n = br.ue();
        if (n > (uint32_t)n_ctb) return "num_entry_point_offsets out of range";
        if (n) { int len = (int)br.ue() + 1; if (len > 32) return "offset_len_minus1 out of range"; for (uint32_t i = 0; i < n; i++) br.u(len); }
    }
    if (pp.sh_extension) { uint32_t n = br.ue(); if (n > 256) return "slice header extension too long"; for (uint32_t i = 0; i < n; i++) br.u(8); }
    if (!br.u1()) return "slice header: alignment bit missing";
    br.align_zero();
    if (br.overrun()) return "slice header truncated";
    sh.data_offset = br.bitpos() >> 3;
    return "";
}

}  // namespace jmamd
