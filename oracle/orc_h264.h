/*
 * oracle/orc_h264.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A scalar, spec-literal restatement of what the reference's closed decode
 * boundary computes: jm_nvdec_decode_frame -> cuvidParseVideoData ->
 * cuvidDecodePicture (/root/reference/nv_dec/nv_dec.cpp:368-403, :33-41)
 * followed by display-order hand-off (nv_dec.cpp:44-52) and the display-area
 * crop of nvdec_create_decoder (nv_dec.cpp:513-519).
 *
 * The arithmetic lives in NVIDIA's closed nvcuvid + NVDEC ASIC (not under
 * /root/reference, nothing to compile or import).  Per ITU-T H.264 a
 * conforming decoder's output is unique, so this oracle restates the
 * normative H.264 decoding process (clauses 7, 8, 9.1, 9.2 and C.4).
 *
 * PARITY STATUS: "parity unpinned" by the reference (it ships no fixtures,
 * golden YUV or checksums -- SURVEY.md section 4 / 8c).  What pins this file
 * instead: I_PCM known-answer streams (decoded samples == payload bytes),
 * VLC-table structural checks (prefix-freeness / Kraft sums), and agreement
 * with an independently written encoder's reconstruction loop
 * (tools/h264gen.c) -- see tests/test_oracle_*.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this code.  The product (jmcodec_amd/csrc) never does.
 */
#ifndef ORC_H264_H
#define ORC_H264_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcDec OrcDec;

/* One output picture, display order, already cropped the way the reference
 * crops: width = crop right-left, height = bottom-top, origin forced to (0,0)
 * (nv_dec.cpp:513-519).  Planes are planar I420 with the given strides. */
typedef struct OrcFrame {
    const uint8_t *y, *u, *v;
    int width, height;     /* display size                              */
    int stride_y, stride_c;
    int poc;               /* picture order count (diagnostic)          */
    int frame_type;        /* 0 = I, 1 = P, 2 = B   (diagnostic)        */
    int decode_index;      /* decode-order index (diagnostic)           */
} OrcFrame;

typedef void (*orc_frame_cb)(void *user, const OrcFrame *f);

OrcDec *orc_open(orc_frame_cb cb, void *user);
void    orc_close(OrcDec *d);

/* Feed one NAL unit WITHOUT start code (EBSP: emulation prevention bytes
 * still present).  Returns 0 on success, <0 on unsupported/corrupt input. */
int orc_decode_nal(OrcDec *d, const uint8_t *nal, size_t len);

/* Feed an Annex-B buffer containing whole NAL units (start codes 00 00 01 /
 * 00 00 00 01); splits and calls orc_decode_nal.  Returns number of NALs
 * consumed or <0 on error. */
int orc_decode_annexb(OrcDec *d, const uint8_t *buf, size_t len);

/* End of stream: finish the current picture and flush the DPB in display
 * order (the cuvid ENDOFSTREAM packet of nv_dec.cpp:389-392). */
void orc_flush(OrcDec *d);

/* Last error string (static storage inside the decoder). */
const char *orc_last_error(const OrcDec *d);
/* tool-usage counters of the stream decoded so far: name of counter i (NULL past the end) and its value */
const char *orc_tool_name(int i);
long orc_tool_count(const OrcDec *d, int i);

/* Stream info valid after the first SPS is activated (nv_dec.cpp:838-845). */
int orc_stream_info(const OrcDec *d, int *disp_w, int *disp_h, int *coded_w, int *coded_h);

/* Convenience for tests: decode a complete Annex-B stream, append every
 * output frame as tight I420 (out_fmt=1) or tight NV12 (out_fmt=0) into a
 * malloc'ed buffer (caller frees with orc_free).  Returns frame count or <0. */
int  orc_decode_stream_to_buffer(const uint8_t *buf, size_t len, int out_fmt,
                                 uint8_t **out, size_t *out_len, int *w, int *h);
void orc_free(void *p);

/* Syntax digest (tests only): FNV-1a over every parsed macroblock's syntax elements. */
void orc_digest_enable(OrcDec *d);
uint64_t orc_digest_value(const OrcDec *d, uint64_t *n_mbs);

/* ---- pack-out restatement (orc_packout.c) ------------------------------ */
/* Byte-for-byte restatement of jm_nvdec_output_frame (nv_dec.cpp:750-828):
 * src is pitch-linear NV12 (luma pitch*height, then interleaved UV rows at
 * src + pitch*height), dst is tight.  out_fmt 0 = NV12, 1 = "YV12" which the
 * reference actually writes as Y,U,V (I420 order, nv_dec.cpp:812-818).
 * *out_len: capacity in, bytes out.  Returns size, -1 no frame, -2 too small. */
int orc_packout(const uint8_t *src, int pitch, int width, int height, int out_fmt,
                uint8_t *dst, int *out_len);

#ifdef __cplusplus
}
#endif
#endif
