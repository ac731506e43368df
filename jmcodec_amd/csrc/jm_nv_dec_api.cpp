// jmcodec_amd/csrc/jm_nv_dec_api.cpp -- the drop-in symbols.
//
// The reference header /root/reference/nv_dec/jm_nv_dec.h:20-88 declares ten free functions with
// C++ linkage (no extern "C", bool in three signatures).  A program compiled against that header
// (test_nv_dec/test_nv_dec.cpp:163-259, test_player/test_player.cpp:206-323) therefore needs the
// Itanium-mangled names below; the declarations here repeat those signatures exactly and forward
// to the C ABI.
#include "../../include/jm_amd_dec.h"

#define JM_EXPORT __attribute__((visibility("default")))
typedef void *handle_nvdec;                                                                     // jm_nv_dec.h:20

JM_EXPORT handle_nvdec jm_nvdec_create_handle() { return jm_amddec_create_handle(); }           // _Z22jm_nvdec_create_handlev
JM_EXPORT int jm_nvdec_init(int codec_type, int out_fmt, char *extra_data, int len, handle_nvdec handle) {   // _Z13jm_nvdec_initiiPciPv
    return jm_amddec_init(codec_type, out_fmt, extra_data, len, handle);
}
JM_EXPORT int jm_nvdec_deinit(handle_nvdec handle) { return jm_amddec_deinit(handle); }         // _Z15jm_nvdec_deinitPv
JM_EXPORT int jm_nvdec_decode_frame(unsigned char *in_buf, int in_data_len, int *got_frame, handle_nvdec handle) {   // _Z21jm_nvdec_decode_framePhiPiPv
    return jm_amddec_decode_frame(in_buf, in_data_len, got_frame, handle);
}
JM_EXPORT int jm_nvdec_output_frame(unsigned char *out_buf, int *out_len, handle_nvdec handle) {  // _Z21jm_nvdec_output_framePhPiPv
    return jm_amddec_output_frame(out_buf, out_len, handle);
}
JM_EXPORT int jm_nvdec_stream_info(int *disp_width, int *disp_height, handle_nvdec handle) {    // _Z20jm_nvdec_stream_infoPiS_Pv
    return jm_amddec_stream_info(disp_width, disp_height, handle);
}
JM_EXPORT void jm_nvdec_set_eof(bool is_eof, handle_nvdec handle) { jm_amddec_set_eof(is_eof ? 1 : 0, handle); }   // _Z16jm_nvdec_set_eofbPv
JM_EXPORT bool jm_nvdec_is_exit(handle_nvdec handle) { return jm_amddec_is_exit(handle) != 0; } // _Z16jm_nvdec_is_exitPv
JM_EXPORT char *jm_nvdec_show_dec_info(handle_nvdec handle) { return jm_amddec_show_dec_info(handle); }   // _Z22jm_nvdec_show_dec_infoPv
JM_EXPORT bool jm_nvdec_is_hw_support() { return jm_amddec_is_hw_support() != 0; }              // _Z22jm_nvdec_is_hw_supportv
