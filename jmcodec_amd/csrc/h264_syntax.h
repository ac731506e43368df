// jmcodec_amd/csrc/h264_syntax.h -- parameter sets and slice header (H.264 7.3.2.1, 7.3.2.2, 7.3.3).
// Host part of the replacement for cuvidParseVideoData (/root/reference/nv_dec/nv_dec.cpp:394);
// the sequence information it yields plays the role of CUVIDEOFORMAT in
// cuvid_handle_video_sequence (nv_dec.cpp:23-30, :496-540).
#pragma once
#include "bitreader.h"
#include <string>

namespace jmamd {

struct SeqParams {
    bool valid = false;
    int profile_idc = 0, constraint_flags = 0, level_idc = 0, id = 0;
    int chroma_format_idc = 1, bit_depth_luma = 8, bit_depth_chroma = 8;
    bool scaling_matrix_present = false;
    uint8_t scaling4[6][16], scaling8[2][64];      // zig-zag order, as transmitted
    int log2_max_frame_num = 4, poc_type = 0, log2_max_poc_lsb = 4;
    bool delta_pic_order_always_zero = false;
    int offset_for_non_ref_pic = 0, offset_for_top_to_bottom = 0, num_ref_frames_in_poc_cycle = 0;
    int offset_for_ref_frame[256];
    int max_num_ref_frames = 0;
    bool gaps_allowed = false;
    int mb_w = 0, mb_h = 0;
    bool frame_mbs_only = true, mbaff = false, direct_8x8_inference = false;
    int crop_l = 0, crop_r = 0, crop_t = 0, crop_b = 0;
    int max_num_reorder_frames = -1, max_dec_frame_buffering = -1;
    // VUI timing_info (E.2.1): a field lasts num_units_in_tick / time_scale seconds, so the frame rate is time_scale / (2 * num_units_in_tick) --
    // what Media SDK's DecodeHeader puts into FrameRateExtN / FrameRateExtD and the reference divides (intel_dec.cpp:975-990).  0 / 0 = not transmitted.
    uint32_t num_units_in_tick = 0, time_scale = 0; bool fixed_frame_rate = false;
    int coded_w() const { return mb_w * 16; }
    int coded_h() const { return mb_h * 16; }
    // nv_dec.cpp:513-519: target size = display_area right-left x bottom-top (origin forced to 0,0)
    int disp_w() const { int w = coded_w() - 2 * (crop_l + crop_r); return w > 0 ? w : coded_w(); }
    int disp_h() const { int h = coded_h() - 2 * (crop_t + crop_b); return h > 0 ? h : coded_h(); }
    int dpb_frames() const;
};

struct PicParamSet {
    bool valid = false;
    int id = 0, sps_id = 0;
    bool cabac = false, bottom_field_poc_present = false;
    int num_ref_idx_default[2] = {1, 1};
    bool weighted_pred = false; int weighted_bipred_idc = 0;
    int init_qp = 26, chroma_qp_off = 0, second_chroma_qp_off = 0;
    bool deblock_ctrl_present = false, constrained_intra = false, redundant_pic_cnt_present = false;
    bool transform8x8 = false, scaling_matrix_present = false;
    uint8_t scaling4[6][16], scaling8[2][64];
};

enum { SL_P = 0, SL_B = 1, SL_I = 2 };

struct RefMod { uint8_t idc; uint32_t val; };
struct MarkOp { uint8_t op; uint32_t a, b; };      // a: diff_pic_nums_minus1 / long_term_pic_num / max_idx_plus1 ; b: long_term_frame_idx

struct SliceHeader {
    int nal_ref_idc = 0; bool idr = false;
    int first_mb = 0, type = SL_I, pps_id = 0, frame_num = 0, idr_pic_id = 0;
    int redundant_pic_cnt = 0;                          // > 0: a slice of a redundant coded picture (7.4.3): dropped (decoder.cpp handle_nal)
    bool field_pic = false, bottom_field = false;      // the picture is ONE FIELD of a frame (PAFF), decoded as a picture of half the height
    int poc_lsb = 0, delta_poc_bottom = 0, delta_poc[2] = {0, 0};
    int num_ref_idx[2] = {0, 0};
    int n_mod[2] = {0, 0}; RefMod mod[2][66];
    bool explicit_wp = false, wp_nondefault = false; int luma_log2_wd = 0, chroma_log2_wd = 0;
    int16_t luma_w[2][32], luma_o[2][32], chroma_w[2][32][2], chroma_o[2][32][2];     // [list][ref_idx]
    bool long_term_reference = false, adaptive_marking = false; int n_mark = 0; MarkOp mark[66];
    int cabac_init_idc = 0; bool direct_spatial_mv_pred = false;
    int qp = 26, disable_deblock = 0, alpha_off = 0, beta_off = 0;     // offsets already doubled
    size_t data_bit_offset = 0;                                        // first bit of slice_data()
};

struct ParamSets {
    SeqParams sps[32];
    PicParamSet pps[256];
    // returns "" on success, else a reason
    std::string parse_sps(BitReader &br);
    std::string parse_pps(BitReader &br);
    std::string parse_slice_header(BitReader &br, int nal_type, int nal_ref_idc, SliceHeader &sh) const;
};

}  // namespace jmamd
