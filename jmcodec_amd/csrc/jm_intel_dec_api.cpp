// jmcodec_amd/csrc/jm_intel_dec_api.cpp -- push/pull decode API of /root/reference/intel_dec/jm_intel_dec.h over the MI355X engine
// (BASELINE config 0's call shape, test_intel_dec/test_intel_dec.cpp:64-102).  The reference runs a worker thread that pulls from an
// input bitstream buffer and queues decoded surfaces (intel_dec.cpp:46-81, 189-332); here the engine already is asynchronous and the decoder already
// keeps its display frames in device memory in display order, so the facade only adapts the call protocol and never holds a frame itself:
//   input_data   = jm_amddec_push_data (the whole chunk is parsed and dispatched at once; a frame the caller has not fetched yet stays current);
//   output_frame = take the next finished display frame and let the decoder put it into the CALLER'S buffer with its one copy-engine transfer --
//                  one copy per frame, the same as jm_nvdec_output_frame (round 5 had three: device -> a scratch buffer -> a fresh vector -> the caller);
//   need_more_data is true while the decoder runs low on work -- fewer than kPicturesLow pictures on their way through parse and device -- and fewer than
//                  kFramesHigh display frames wait for the caller: a caller following test_intel_dec.cpp:78-102 (one output_frame per loop turn, up to
//                  free_buf_len bytes of input per turn -- some 45 pictures of the 1080p C1 stream) keeps the pipeline fed without running far ahead of its
//                  own consumption; while it is held off, output_frame sleeps up to kWaitUs for the frame that is on its way instead of returning -1 into a
//                  spinning loop.  (A first version looked at the waiting frames only, 16 of them: the caller then pushed when its OUTPUT ran low, which is
//                  when the pipeline had long run dry -- 0.80 of the NAL-per-call rate.)
#include "../../include/jm_amd_dec.h"
#include "../../include/jm_amd_intel_dec.h"
#include <mutex>
#include <vector>
#include <cstring>

#define JM_EXPORT __attribute__((visibility("default")))

namespace {
constexpr int kInputChunk = 1 << 20;      // what free_buf_len reports (the reference's buffer starts at 1 MB, intel_dec.h)
constexpr long long kFramesHigh = 64;     // display frames waiting for the caller: more than that and input is held off whatever the pipeline holds
constexpr long long kPicturesLow = 24;    // pictures on their way (parse, engine, device): fewer than that and the decoder wants input
constexpr int kWaitUs = 2000;             // longest sleep of output_frame for a frame that is being decoded, while input is held off
struct Ctx {
    jm_amddec_handle dec = nullptr;
    std::mutex m;
    jm_amdintel_yuv_callback cb = nullptr; void *user = nullptr;
    std::vector<unsigned char> cb_buf;    // callback mode: the one buffer every frame is delivered in
    bool eof = false, inited = false;
    bool have = false;                    // the decoder holds a current frame that nobody fetched yet
    long long waiting() { return jm_amddec_get_stat(dec, "frames_waiting") + (have ? 1 : 0); }
    bool held_off() { return !cb && (waiting() >= kFramesHigh || jm_amddec_get_stat(dec, "pictures_in_flight") >= kPicturesLow); }
    // make the next finished display frame current (m held).  After set_eof the decoder drains: it blocks until the next frame is there or none is left.
    bool take(int wait_us) {
        if (have || !inited) return have;
        int got = 0;
        if (eof) jm_amddec_decode_frame(nullptr, 0, &got, dec);
        else if (wait_us > 0) jm_amddec_wait_frame(&got, wait_us, dec);
        else jm_amddec_poll_frame(&got, dec);
        have = got == 1;
        return have;
    }
    // callback mode: hand every finished frame to the callback (the reference stores the callback and never calls it, intel_dec.cpp:369-376)
    void deliver() {
        while (cb && take(0)) {
            int w = 0, h = 0; jm_amddec_stream_info(&w, &h, dec);
            const size_t cap = (size_t)w * h * 3 / 2 + 16;
            if (cb_buf.size() < cap) cb_buf.resize(cap);
            int n = (int)cb_buf.size();
            have = false;
            if (jm_amddec_output_frame(cb_buf.data(), &n, dec) > 0) cb(cb_buf.data(), n, user);
        }
    }
};
inline Ctx *C(void *h) { return static_cast<Ctx *>(h); }
}  // namespace

extern "C" {
JM_EXPORT jm_amdintel_handle jm_amdintel_create_handle(void) { Ctx *c = new Ctx(); c->dec = jm_amddec_create_handle(); return c; }
JM_EXPORT int jm_amdintel_init(int codec_type, int out_fmt, jm_amdintel_handle h) {
    if (!h) return -1;
    int rc = jm_amddec_init(codec_type, out_fmt, nullptr, 0, C(h)->dec);
    C(h)->inited = rc == 0;
    return rc;
}
JM_EXPORT int jm_amdintel_deinit(jm_amdintel_handle h) { if (!h) return -1; jm_amddec_deinit(C(h)->dec); delete C(h); return 0; }
JM_EXPORT int jm_amdintel_set_yuv_callback(void *user, jm_amdintel_yuv_callback cb, jm_amdintel_handle h) {
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(C(h)->m); C(h)->cb = cb; C(h)->user = user; return 0;
}
JM_EXPORT int jm_amdintel_input_data(unsigned char *in_buf, int n, jm_amdintel_handle h) {
    if (!h || !C(h)->inited || !in_buf || n <= 0 || C(h)->eof) return -1;
    std::lock_guard<std::mutex> lk(C(h)->m);
    if (jm_amddec_push_data(in_buf, n, C(h)->dec) != 0) return -1;
    C(h)->deliver();
    return n;
}
JM_EXPORT int jm_amdintel_output_frame(unsigned char *out_buf, int *out_len, jm_amdintel_handle h) {
    if (!h || !out_len) return -1;
    Ctx *c = C(h);
    std::lock_guard<std::mutex> lk(c->m);
    if (c->cb) { c->deliver(); *out_len = 0; return -1; }                 // frames go to the callback
    if (!c->take(c->held_off() ? kWaitUs : 0)) { *out_len = 0; return -1; }   // intel_dec.cpp:251-255
    int w = 0, hh = 0; jm_amddec_stream_info(&w, &hh, c->dec);
    const int need = w * hh * 3 / 2;
    if (!out_buf) { *out_len = need; return 0; }                          // size query (jm_intel_dec.h:74); the frame stays current
    if (*out_len < need) { *out_len = 0; return -2; }                     // intel_dec.cpp:266-270; the frame stays current
    int n = *out_len;
    c->have = false;
    if (jm_amddec_output_frame(out_buf, &n, c->dec) <= 0) { *out_len = 0; return -1; }
    *out_len = n;
    return 0;
}
JM_EXPORT int jm_amdintel_set_eof(int is_eof, jm_amdintel_handle h) {
    if (!h || !is_eof || !C(h)->inited) return h ? 0 : -1;
    std::lock_guard<std::mutex> lk(C(h)->m);
    // The frame that is current stays current (the decoder's end-of-stream call would take the next one); the end of the stream is sent by the first
    // take() after it (jm_amddec_decode_frame(NULL, 0) = flush, then one display frame per call).
    C(h)->eof = true;
    C(h)->deliver();
    return 0;
}
JM_EXPORT char *jm_amdintel_info(jm_amdintel_handle h) { return h ? jm_amddec_show_dec_info(C(h)->dec) : nullptr; }
// dec_get_stream_info (intel_dec.cpp:975-990): FrameRateExtN / FrameRateExtD, which Media SDK's DecodeHeader takes from the VUI timing information
// (H.264 E.2.1: time_scale / (2 * num_units_in_tick); HEVC E.3.1: vui_time_scale / vui_num_units_in_tick).  A stream without it reports 0.
JM_EXPORT int jm_amdintel_get_stream_info(int *w, int *hh, float *fps, jm_amdintel_handle h) {
    if (!h || !w || !hh) return -1;
    if (fps) {
        const long long num = jm_amddec_get_stat(C(h)->dec, "fps_num"), den = jm_amddec_get_stat(C(h)->dec, "fps_den");
        *fps = num > 0 && den > 0 ? (float)((double)num / (double)den) : 0.0f;
    }
    return jm_amddec_stream_info(w, hh, C(h)->dec);
}
JM_EXPORT int jm_amdintel_need_more_data(jm_amdintel_handle h) {
    if (!h) return 0;
    std::lock_guard<std::mutex> lk(C(h)->m);
    return !C(h)->eof && !C(h)->held_off();
}
JM_EXPORT int jm_amdintel_free_buf_len(jm_amdintel_handle h) { return h ? kInputChunk : 0; }
JM_EXPORT int jm_amdintel_is_exit(jm_amdintel_handle h) {
    if (!h) return 1;
    Ctx *c = C(h);
    std::lock_guard<std::mutex> lk(c->m);
    if (!c->eof) return 0;
    if (c->cb) c->deliver();
    else if (!c->have && !jm_amddec_is_exit(c->dec)) c->take(0);          // the drain call that finds the queue empty is what ends the decoder (nv_dec.cpp:460-466)
    return !c->have && jm_amddec_is_exit(c->dec);
}
JM_EXPORT int jm_amdintel_is_hw_support(void) { return jm_amddec_is_hw_support(); }

JM_EXPORT jm_amddec_handle jm_amdintel_decoder(jm_amdintel_handle h) { return h ? C(h)->dec : nullptr; }

// The loop of test_intel_dec.cpp:64-102 in native code, for callers in interpreted languages (bench.py measures the library, not its own per-call
// overhead): while !is_exit { if need_more_data and input is left: input_data(up to free_buf_len bytes) -- or set_eof when it ran out; output_frame }.
JM_EXPORT long jm_amdintel_run_pushpull(const unsigned char *buf, long len, unsigned char *out_buf, int out_cap, jm_amdintel_handle h) {
    if (!h || !buf || len <= 0 || !out_buf) return -1;
    long pos = 0, frames = 0; bool sent_eof = false;
    while (!jm_amdintel_is_exit(h)) {
        if (!sent_eof && jm_amdintel_need_more_data(h)) {
            const long room = jm_amdintel_free_buf_len(h), n = len - pos < room ? len - pos : room;
            if (n == 0) { sent_eof = true; jm_amdintel_set_eof(1, h); }
            else { if (jm_amdintel_input_data(const_cast<unsigned char *>(buf + pos), (int)n, h) < 0) return -2; pos += n; }
        }
        int n = out_cap;
        if (jm_amdintel_output_frame(out_buf, &n, h) == 0) frames++;
    }
    return frames;
}
}  // extern "C"

// ---- the reference header's own (C++-linkage) names: jm_intel_dec.h:29-122 ----
typedef void *handle_inteldec;
typedef int (*HANDLE_YUV_CALLBACK)(unsigned char *out_buf, int out_len, void *user_data);
JM_EXPORT handle_inteldec jm_intel_dec_create_handle() { return jm_amdintel_create_handle(); }
JM_EXPORT int jm_intel_dec_init(int codec_type, int out_fmt, handle_inteldec handle) { return jm_amdintel_init(codec_type, out_fmt, handle); }
JM_EXPORT int jm_intel_dec_deinit(handle_inteldec handle) { return jm_amdintel_deinit(handle); }
JM_EXPORT int jm_intel_dec_set_yuv_callback(void *user_data, HANDLE_YUV_CALLBACK callback, handle_inteldec handle) {
    return jm_amdintel_set_yuv_callback(user_data, callback, handle); }
JM_EXPORT int jm_intel_dec_input_data(unsigned char *in_buf, int in_data_len, handle_inteldec handle) {
    return jm_amdintel_input_data(in_buf, in_data_len, handle); }
JM_EXPORT int jm_intel_dec_output_frame(unsigned char *out_buf, int *out_len, handle_inteldec handle) {
    return jm_amdintel_output_frame(out_buf, out_len, handle); }
JM_EXPORT int jm_intel_dec_set_eof(int is_eof, handle_inteldec handle) { return jm_amdintel_set_eof(is_eof, handle); }
JM_EXPORT char *jm_intel_dec_info(handle_inteldec handle) { return jm_amdintel_info(handle); }
JM_EXPORT int jm_intel_get_stream_info(int *width, int *height, float *frame_rate, handle_inteldec handle) {
    return jm_amdintel_get_stream_info(width, height, frame_rate, handle); }
JM_EXPORT bool jm_intel_dec_need_more_data(handle_inteldec handle) { return jm_amdintel_need_more_data(handle) != 0; }
JM_EXPORT int jm_intel_dec_free_buf_len(handle_inteldec handle) { return jm_amdintel_free_buf_len(handle); }
JM_EXPORT bool jm_intel_dec_is_exit(handle_inteldec handle) { return jm_amdintel_is_exit(handle) != 0; }
JM_EXPORT bool jm_intel_is_hw_support() { return jm_amdintel_is_hw_support() != 0; }
