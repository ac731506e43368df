/*
 * include/jm_amd_dec.h -- C ABI of the MI355X-native decode backend (libjm_amd_dec.so).
 *
 * Every entry point mirrors one function of the reference's public decode API
 * (/root/reference/nv_dec/jm_nv_dec.h) 1:1 -- same argument order, meaning and
 * return conventions -- with plain C types so that any FFI can bind it.  The
 * same library ALSO exports the reference's own C++-mangled jm_nvdec_* symbols
 * (jmcodec_amd/csrc/jm_nv_dec_api.cpp) so that test_nv_dec.cpp links unchanged.
 *
 *   jm_amddec_create_handle   <- jm_nvdec_create_handle   jm_nv_dec.h:27   (nv_dec.cpp:695-698)
 *   jm_amddec_init            <- jm_nvdec_init            jm_nv_dec.h:39   (nv_dec.cpp:710-713)
 *   jm_amddec_deinit          <- jm_nvdec_deinit          jm_nv_dec.h:47   (nv_dec.cpp:721-724)
 *   jm_amddec_decode_frame    <- jm_nvdec_decode_frame    jm_nv_dec.h:58   (nv_dec.cpp:735-739)
 *   jm_amddec_output_frame    <- jm_nvdec_output_frame    jm_nv_dec.h:68   (nv_dec.cpp:750-828)
 *   jm_amddec_stream_info     <- jm_nvdec_stream_info     jm_nv_dec.h:79   (nv_dec.cpp:838-845)
 *   jm_amddec_set_eof         <- jm_nvdec_set_eof         jm_nv_dec.h:82   (nv_dec.cpp:848-851)
 *   jm_amddec_is_exit         <- jm_nvdec_is_exit         jm_nv_dec.h:84   (nv_dec.cpp:853-856)
 *   jm_amddec_show_dec_info   <- jm_nvdec_show_dec_info   jm_nv_dec.h:86   (nv_dec.cpp:858-861)
 *   jm_amddec_is_hw_support   <- jm_nvdec_is_hw_support   jm_nv_dec.h:88   (nv_dec.cpp:863-870)
 *
 * Differences from the reference, all deliberate:
 *   - init/decode_frame return -1 (and log to stderr) when no HIP device can be
 *     used or the stream is unsupported; the reference returns 0 unconditionally
 *     (nv_dec.cpp:79,493).  There is no CPU fallback.
 *   - handles are independent: N handles may be driven from N threads.
 *   - the device is chosen by JM_AMD_DEC_DEVICE, jm_amddec_set_option("device")
 *     before init, or round-robin over visible devices (reference: device 0,
 *     nv_dec.cpp:209).
 */
#ifndef JM_AMD_DEC_H
#define JM_AMD_DEC_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void *jm_amddec_handle;

jm_amddec_handle jm_amddec_create_handle(void);
/* codec_type: 0 = H.264, 1 = H.265 / HEVC Main (enum nv_dec.h:37-46); out_fmt: 0 = NV12, 1 = "YV12" = planar Y,U,V */
int  jm_amddec_init(int codec_type, int out_fmt, char *extra_data, int len, jm_amddec_handle h);
int  jm_amddec_deinit(jm_amddec_handle h);
/* in_buf may hold any chunk of an Annex-B stream; (NULL, 0) signals end of stream and then
 * drains one display-order frame per call.  *got_frame = 1 when a frame is ready. */
int  jm_amddec_decode_frame(unsigned char *in_buf, int in_data_len, int *got_frame, jm_amddec_handle h);
/* *out_len: capacity in, bytes out.  Returns the frame size (>0), -1 no frame, -2 buffer too small. */
int  jm_amddec_output_frame(unsigned char *out_buf, int *out_len, jm_amddec_handle h);
int  jm_amddec_stream_info(int *disp_width, int *disp_height, jm_amddec_handle h);
void jm_amddec_set_eof(int is_eof, jm_amddec_handle h);
int  jm_amddec_is_exit(jm_amddec_handle h);
char *jm_amddec_show_dec_info(jm_amddec_handle h);
int  jm_amddec_is_hw_support(void);

/* ---- additions (no reference counterpart) ---- */
/* keys: "device" (before init), "device_output" (before init, see below), "sync" (1 = every call waits for the pipeline; deterministic),
 *       "parse_only" (1 = host bitstream stages only, frames carry no pixels; for host-side tests),
 *       "digest" (1 = accumulate the macroblock syntax digest; implies sync),
 *       "display_delay" (n: a frame is handed out only while n pictures of the handle are still on their way -- the reference's ulMaxDisplayDelay,
 *       nv_dec.cpp:341; default 0), "profile" (1 = time the kernels with events), "wait_idle" (block until every dispatched picture has run),
 *       engine-wide after init: "chain_depth" (pictures of one stream per chain launch; 0 = the defaults: 8, and 16 while one or two streams are active),
 *       "chain_lag", "chain_streams", "debug_stall" (DESIGN.md 4b); before init: "job_slots" (pictures in flight per handle, 8..64; default 40 for H.264 up
 *       to 1080p, else 24);
 *       tests only: "fast_parse" (0 = every macroblock through the general parser path), "job_digest" (1 = digest of the job lists; implies sync) */
/* like jm_amddec_decode_frame without input: *got_frame = 1 when a display-order frame became ready (never signals end of stream) */
int  jm_amddec_poll_frame(int *got_frame, jm_amddec_handle h);
/* jm_amddec_poll_frame that sleeps up to timeout_us microseconds for a frame whose picture is still being decoded (returns at once when nothing is on its way) */
int  jm_amddec_wait_frame(int *got_frame, int timeout_us, jm_amddec_handle h);
/* input without taking a frame: the push half of the reference's push / pull API (intel_dec_put_input_data, /root/reference/intel_dec/intel_dec.cpp:189-234).
 * The frame signalled by an earlier got_frame = 1 stays current until the next jm_amddec_decode_frame / jm_amddec_poll_frame call.  0, or -1 on error. */
int  jm_amddec_push_data(unsigned char *in_buf, int in_data_len, jm_amddec_handle h);
/* end of stream without taking a frame (what jm_amddec_decode_frame(NULL, 0) does before it hands out a frame); drain with jm_amddec_decode_frame(NULL, 0)
 * afterwards.  jm_amddec_push_data / jm_amddec_push_eos may run on another thread than jm_amddec_poll_frame / _wait_frame / _output_frame: the facade of
 * the push / pull API feeds from a worker thread, as the reference's does (intel_dec.cpp:46-81) */
int  jm_amddec_push_eos(jm_amddec_handle h);
int  jm_amddec_set_option(jm_amddec_handle h, const char *key, long long value);
/* keys: "frames", "pictures", "job_bytes", "errors", "intra_mbs", "coef_int16", "syntax_digest",
 *       "digest_mbs", "i_pictures", "p_pictures", "coded_width", "coded_height", "pitch", "device",
 *       "threads", "elapsed_us", "display_poc:<n>", "fps_num" / "fps_den" (frame rate from the VUI timing information, 0 / 0 = not transmitted),
 *       "frames_waiting" (display frames decided and not yet made current by a decode / poll call), "frames_done_unfetched" (those of them whose samples are there), "device_wait_errors", "direct_frames" / "direct_ns" (frames that left by one copy-engine
 *       transfer into the caller's buffer, and the time their callers waited), "copy_engines" (SDMA engines used for that, bit mask),
 *       "job_digest", "eng_*" / "k_*" (engine and per-kernel counters, bench.py) */
long long jm_amddec_get_stat(jm_amddec_handle h, const char *key);
const char *jm_amddec_last_error(jm_amddec_handle h);

/* Stand-alone pack-out of one pitch-linear NV12 surface that already lives in device memory
 * (device pointers).  Same semantics as jm_nvdec_output_frame's repack (nv_dec.cpp:782-820).
 * stream: a hipStream_t or NULL.  Returns 0 or a negative hipError. */
/* SURVEY 8f f3 -- device-resident output, the path the reference stubbed out (nv_dec.h:98-107, nv_dec.cpp:244-265 under "#if 0").
 * With option "device_output" = 1 (before init) display frames are not copied to the host at all:
 *   jm_amddec_output_frame_device: *dev = device pointer of the current frame (tight NV12 / I420 as chosen at init), *len = its
 *     size; valid until the next jm_amddec_decode_frame call.  Works in the default mode too (the staging copy of the frame).
 *   jm_amddec_output_argb_device: converts the current frame to ARGB32 (memory bytes B,G,R,A; BT.601 limited range) into a device
 *     buffer of `pitch` bytes per row (>= 4 * width).  Returns 0, or -1 when no frame is current. */
int  jm_amddec_output_frame_device(void **dev, int *len, jm_amddec_handle h);
int  jm_amddec_output_argb_device(void *dev_dst, int pitch, jm_amddec_handle h);
int  jm_amddec_packout_device(const void *d_src, int pitch, int width, int height, int out_fmt,
                              void *d_dst, void *stream);
/* SURVEY 8f f4 -- the encoder-side pre-processing of the reference (/root/reference/nv_enc/nv_enc.cpp:1022-1079: cuMemcpy2D of the luma plane +
 * the InterleaveUV kernel; the CPU loop of intel_enc.cpp:316-387) as one HIP kernel, device to device: a tight frame (src_fmt 1 = I420
 * planar Y,U,V; 0 = tight NV12) becomes a pitch-linear NV12 surface (luma rows at `pitch`, interleaved UV rows from row `height`), the layout an
 * encoder's input surface has.  width and height must be even, pitch >= width.
 *   jm_amddec_i420_to_nv12_device: stand-alone (any device frame).  stream: a hipStream_t or NULL.  Returns 0 or -1.
 *   jm_amddec_output_nv12_pitch_device: the same for the decoder's current display frame -- decode -> encoder surface without touching the host
 *     (the first half of the transcode loop the reference's README leaves unfinished).  Returns 0, or -1 when no frame is current. */
int  jm_amddec_i420_to_nv12_device(const void *d_src, int width, int height, int src_fmt, void *d_dst, int pitch, void *stream);
int  jm_amddec_output_nv12_pitch_device(void *dev_dst, int pitch, jm_amddec_handle h);

/* The hot loop of the reference harness in native code (/root/reference/test_nv_dec/test_nv_dec.cpp:184-250): feed the Annex-B buffer one
 * NAL unit per jm_nvdec_decode_frame call (a NAL = start code + payload up to the next start code, :63-86), fetch a frame with
 * jm_nvdec_output_frame into out_buf whenever got_frame == 1.  `passes` repeats the buffer.  Does not send end of stream (the caller
* decides when to drain).  out_buf == NULL (with option "device_output"): frames are taken with jm_amddec_output_frame_device instead, i.e. they
 * stay in device memory and nothing crosses PCIe (profiling: rocprofv3 replaces copy-engine transfers by blit kernels that disturb the decode
 * kernels).  Returns the number of frames fetched, < 0 on error.  Exists so that callers in interpreted languages
 * (bench.py) measure the library, not their own per-call overhead. */
long jm_amddec_feed_annexb(const unsigned char *buf, long len, int passes, unsigned char *out_buf, int out_cap, jm_amddec_handle handle);

#ifdef __cplusplus
}
#endif
#endif
