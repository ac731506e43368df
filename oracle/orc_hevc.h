/*
 * oracle/orc_hevc.h -- CPU ORACLE for the HEVC leg (test infrastructure, NOT the product).
 *
 * A scalar, spec-literal restatement of what the reference's closed decode boundary computes when
 * jm_nvdec_init is called with codec_type 1 (/root/reference/nv_dec/nv_dec.h:37-46 enum, nv_dec.cpp:629-661 maps it
 * to cudaVideoCodec_HEVC): cuvidParseVideoData -> cuvidDecodePicture (nv_dec.cpp:368-403, :33-41,
 * CUVIDHEVCPICPARAMS nv_sdk/inc/dynlink_cuviddec.h:428-530), display-order hand-off (:44-52) and the display-area crop
 * (:513-519).  The arithmetic lives in NVIDIA's closed nvcuvid + NVDEC ASIC; ITU-T H.265 defines a conforming decoder's
 * output uniquely, so this file set restates the normative decoding process (clauses 7, 8, 9.3 and C.5.2) for the
 * Main profile (8-bit 4:2:0).
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no HEVC fixture, golden YUV or checksum, and this
 * image holds no third-party HEVC stream or decoder either, so every table in orc_hevc_tables.h (CABAC context
 * initialisation values, transform matrix, filters, beta/tC) is restated from the published standard without an external
 * check.  What pins it instead: agreement with an independently written encoder's reconstruction loop (tools/hevcgen.c)
 * and structural tests on the tables (tests/test_hevc_*.py), a pcm_sample known answer (all-PCM 8-bit stream: decoded blocks
 * == payload bytes, tests/test_hevc_oracle.py::test_pcm_known_answer) and a separately typed copy of the 462 initValues
 * (tests/test_table_provenance.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this code.
 */
#ifndef ORC_HEVC_H
#define ORC_HEVC_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrchDec OrchDec;
typedef struct OrchFrame {
    const uint8_t *y, *u, *v;      /* planar, already cropped to the conformance window with origin (0,0) */
    int width, height, stride_y, stride_c;
    int poc, slice_type, decode_index;
} OrchFrame;
typedef void (*orch_frame_cb)(void *user, const OrchFrame *f);

OrchDec *orch_open(orch_frame_cb cb, void *user);
void orch_close(OrchDec *d);
int  orch_decode_nal(OrchDec *d, const uint8_t *nal, size_t len);          /* one NAL unit without start code */
int  orch_decode_annexb(OrchDec *d, const uint8_t *buf, size_t len);
void orch_flush(OrchDec *d);
const char *orch_last_error(const OrchDec *d);
int  orch_stream_info(const OrchDec *d, int *disp_w, int *disp_h, int *coded_w, int *coded_h);
/* decode a whole Annex-B stream; frames appended as tight I420 (out_fmt 1) or NV12 (0); returns frame count or < 0 */
int  orch_decode_stream_to_buffer(const uint8_t *buf, size_t len, int out_fmt, uint8_t **out, size_t *out_len, int *w, int *h);
void orch_free(void *p);
/* syntax digest (tests only): FNV-1a over a canonical record of every coding unit, in decoding order */
void orch_digest_enable(OrchDec *d);
uint64_t orch_digest_value(const OrchDec *d, uint64_t *n_cus);
/* tool-usage counters: name of counter i (NULL past the end) and its value */
const char *orch_tool_name(int i);
long orch_tool_count(const OrchDec *d, int i);

#ifdef __cplusplus
}
#endif
#endif
