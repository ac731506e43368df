// jmcodec_amd/csrc/hevc_syntax.h -- HEVC parameter sets and slice segment header (ITU-T H.265 7.3.2.2, 7.3.2.3, 7.3.6).
// Host part of the replacement for cuvidParseVideoData with codec_type 1 (/root/reference/nv_dec/nv_dec.cpp:394, codec enum
// nv_dec.h:37-46); what it yields plays the role of CUVIDEOFORMAT / CUVIDHEVCPICPARAMS (nv_sdk/inc/dynlink_cuviddec.h:428-530).
// Main profile only (8-bit 4:2:0).
#pragma once
#include "bitreader.h"
#include <string>

namespace jmamd {

struct HevcRps {                       // short-term reference picture set: deltas sorted closest first
    int n_neg = 0, n_pos = 0;
    int16_t d[2][16]; uint8_t used[2][16];
};

struct HevcScaling {                   // ScalingFactor m[x][y] as [y * n + x] (7.4.5)
    uint8_t f4[6][16], f8[6][64], f16[6][256], f32[2][1024];
    void set_default();
};

struct HevcSps {
    bool valid = false;
    int width = 0, height = 0, conf[4] = {0, 0, 0, 0};
    int log2_max_poc_lsb = 4, max_sub_layers = 1;
    int max_dec_pic_buffering = 1, max_num_reorder = 0;
    int log2_min_cb = 3, log2_ctb = 4, log2_min_tb = 2, log2_max_tb = 5, depth_inter = 0, depth_intra = 0;
    bool scaling_enabled = false, scaling_present = false; HevcScaling sf;
    bool amp = false, sao = false, pcm = false, pcm_loop_filter_disabled = false;
    int pcm_bits_y = 8, pcm_bits_c = 8, log2_min_pcm = 3, log2_max_pcm = 3;
    int n_rps = 0; HevcRps rps[65];
    bool long_term_present = false; int n_lt = 0; uint16_t lt_lsb[32]; uint8_t lt_used[32];
    bool temporal_mvp = false, strong_intra = false;
    uint32_t num_units_in_tick = 0, time_scale = 0;    // vui_timing_info (E.3.1): a picture lasts num_units_in_tick / time_scale seconds; 0 = not transmitted
    // nv_dec.cpp:513-519: target size = display area size, origin forced to (0,0)
    int disp_w() const { return width - 2 * (conf[0] + conf[1]); }
    int disp_h() const { return height - 2 * (conf[2] + conf[3]); }
};

struct HevcPps {
    bool valid = false;
    int sps_id = 0;
    bool dependent_slices = false, output_flag_present = false, sign_hiding = false, cabac_init_present = false;
    int n_extra_bits = 0, n_ref_default[2] = {1, 1}, init_qp = 26;
    bool constrained_intra = false, transform_skip = false, cu_qp_delta = false;
    int diff_cu_qp_delta_depth = 0, cb_qp_off = 0, cr_qp_off = 0;
    bool slice_chroma_qp_offsets = false, weighted_pred = false, weighted_bipred = false, tq_bypass = false;
    bool tiles = false, wpp = false, uniform = true, lf_across_tiles = true;
    int tile_cols = 1, tile_rows = 1, col_w[20], row_h[22];
    bool lf_across_slices = false, deblock_override = false, deblock_disabled = false;
    int beta_off = 0, tc_off = 0;
    bool scaling_present = false; HevcScaling sf;
    bool lists_modification = false, sh_extension = false;
    int log2_par_mrg = 2;
};

enum { HSL_B = 0, HSL_P = 1, HSL_I = 2 };

struct HevcSliceHeader {
    bool first_in_pic = false, no_output_of_prior = false, dependent = false;
    int pps_id = 0, segment_addr = 0, slice_addr = 0;
    int type = HSL_I; bool pic_output = true; int poc_lsb = 0;
    HevcRps rps;
    int n_lt = 0; int lt_poc[32]; uint8_t lt_used[32], lt_msb[32];     // lt_poc: lsb, or lsb - cycle * MaxLsb when lt_msb (resolved against the current POC)
    bool temporal_mvp = false, sao_luma = false, sao_chroma = false;
    int n_ref[2] = {0, 0}; bool rplm[2] = {false, false}; uint8_t list_entry[2][16];
    bool mvd_l1_zero = false, cabac_init = false, col_from_l0 = true; int col_ref_idx = 0;
    bool has_wp = false; int wp_denom[2] = {0, 0}; int16_t wp_w[2][16][3], wp_o[2][16][3];
    int max_merge = 5, qp = 26, cb_qp_off = 0, cr_qp_off = 0;
    bool deblock_disabled = false, lf_across_slices = false; int beta_off = 0, tc_off = 0;
    size_t data_offset = 0;            // first byte of slice_segment_data() in the RBSP
};

struct HevcParamSets {
    HevcSps sps[16];
    HevcPps pps[64];
    // "" on success, else the reason
    std::string parse_sps(BitReader &br);
    std::string parse_pps(BitReader &br);
    // `prev`: the preceding slice segment header of the same picture (dependent segments inherit it)
    std::string parse_slice_header(BitReader &br, int nal_type, HevcSliceHeader &sh, const HevcSliceHeader *prev) const;
};

}  // namespace jmamd
