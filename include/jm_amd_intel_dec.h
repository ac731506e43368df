/*
 * include/jm_amd_intel_dec.h -- C ABI of the push/pull decode API.
 *
 * Replaces, entry point by entry point, /root/reference/intel_dec/jm_intel_dec.h:29-121 (implemented there over Intel
 * Media SDK in intel_dec.cpp:189-376, 1022-1127) on top of the same MI355X decode engine as jm_amd_dec.h:
 *   jm_amdintel_create_handle   <- jm_intel_dec_create_handle    jm_intel_dec.h:29
 *   jm_amdintel_init            <- jm_intel_dec_init             jm_intel_dec.h:39
 *   jm_amdintel_deinit          <- jm_intel_dec_deinit           jm_intel_dec.h:47
 *   jm_amdintel_set_yuv_callback<- jm_intel_dec_set_yuv_callback jm_intel_dec.h:57   (the reference stores the callback but
 *                                   never calls it, intel_dec.cpp:369-376; here ready frames ARE delivered to it)
 *   jm_amdintel_input_data      <- jm_intel_dec_input_data       jm_intel_dec.h:67   (intel_dec_put_input_data, :189-234)
 *   jm_amdintel_output_frame    <- jm_intel_dec_output_frame     jm_intel_dec.h:78   (intel_dec_output_yuv_frame, :244-332:
 *                                   0 = frame copied, -1 = none ready (*out_len = 0), -2 = buffer too small (*out_len = 0))
 *   jm_amdintel_set_eof         <- jm_intel_dec_set_eof          jm_intel_dec.h:84
 *   jm_amdintel_info            <- jm_intel_dec_info             jm_intel_dec.h:93   (same text block as nv_dec.cpp:663-683)
 *   jm_amdintel_get_stream_info <- jm_intel_get_stream_info      jm_intel_dec.h:95
 *   jm_amdintel_need_more_data  <- jm_intel_dec_need_more_data   jm_intel_dec.h:103
 *   jm_amdintel_free_buf_len    <- jm_intel_dec_free_buf_len     jm_intel_dec.h:111
 *   jm_amdintel_is_exit         <- jm_intel_dec_is_exit          jm_intel_dec.h:119
 *   jm_amdintel_is_hw_support   <- jm_intel_is_hw_support        jm_intel_dec.h:122
 * The library also exports the Itanium-mangled C++ names of the reference header (jm_intel_dec_api.cpp), so
 * test_intel_dec.cpp:64-102 links unchanged.
 */
#ifndef JM_AMD_INTEL_DEC_H
#define JM_AMD_INTEL_DEC_H
#ifdef __cplusplus
extern "C" {
#endif

typedef void *jm_amdintel_handle;
typedef int (*jm_amdintel_yuv_callback)(unsigned char *out_buf, int out_len, void *user_data);

jm_amdintel_handle jm_amdintel_create_handle(void);
int   jm_amdintel_init(int codec_type, int out_fmt, jm_amdintel_handle h);
int   jm_amdintel_deinit(jm_amdintel_handle h);
int   jm_amdintel_set_yuv_callback(void *user_data, jm_amdintel_yuv_callback cb, jm_amdintel_handle h);
/* any chunk of an Annex-B stream, at most jm_amdintel_free_buf_len() bytes; returns bytes accepted (> 0) or < 0 */
int   jm_amdintel_input_data(unsigned char *in_buf, int in_data_len, jm_amdintel_handle h);
int   jm_amdintel_output_frame(unsigned char *out_buf, int *out_len, jm_amdintel_handle h);
int   jm_amdintel_set_eof(int is_eof, jm_amdintel_handle h);
char *jm_amdintel_info(jm_amdintel_handle h);
int   jm_amdintel_get_stream_info(int *width, int *height, float *frame_rate, jm_amdintel_handle h);
int   jm_amdintel_need_more_data(jm_amdintel_handle h);
int   jm_amdintel_free_buf_len(jm_amdintel_handle h);
int   jm_amdintel_is_exit(jm_amdintel_handle h);
int   jm_amdintel_is_hw_support(void);

/* addition (no reference counterpart): the push / pull loop of /root/reference/test_intel_dec/test_intel_dec.cpp:64-102 in native code over a whole
 * Annex-B buffer -- input_data in chunks of free_buf_len while need_more_data, set_eof when the input ran out, one output_frame into out_buf per
 * turn, until is_exit.  Returns the number of frames fetched, < 0 on error.  For callers in interpreted languages (bench.py). */
/* the jm_amd_dec.h handle behind a push / pull handle: options before init (jm_amddec_set_option), statistics (jm_amddec_get_stat), jm_amddec_last_error */
void *jm_amdintel_decoder(jm_amdintel_handle h);
long  jm_amdintel_run_pushpull(const unsigned char *buf, long len, unsigned char *out_buf, int out_cap, jm_amdintel_handle h);

#ifdef __cplusplus
}
#endif
#endif
