// jmcodec_amd/csrc/jm_intel_dec_api.cpp -- push/pull decode API of /root/reference/intel_dec/jm_intel_dec.h over the MI355X engine
// (SURVEY.md 8f f1).  The reference runs a worker thread that pulls from an input bitstream buffer and queues decoded surfaces
// (intel_dec.cpp:46-81, 189-332); here the engine already is asynchronous, so the facade only adapts the call protocol:
// input is handed to the decoder at once, finished frames are collected into a small FIFO that output_frame / the callback drain,
// and need_more_data turns false while that FIFO is full so that a caller following test_intel_dec.cpp:78-102 cannot run ahead.
#include "../../include/jm_amd_dec.h"
#include "../../include/jm_amd_intel_dec.h"
#include <deque>
#include <mutex>
#include <vector>
#include <cstring>

#define JM_EXPORT __attribute__((visibility("default")))

namespace {
constexpr int kInputChunk = 1 << 20;      // what free_buf_len reports (the reference's buffer starts at 1 MB, intel_dec.h)
constexpr size_t kFifoHigh = 8;           // decoded frames held for the caller before input is throttled
struct Ctx {
    jm_amddec_handle dec = nullptr;
    std::mutex m;
    std::deque<std::vector<unsigned char>> fifo;
    std::vector<unsigned char> tmp;
    jm_amdintel_yuv_callback cb = nullptr; void *user = nullptr;
    bool eof = false, inited = false;
    // move every frame the decoder has ready into the FIFO (or to the callback); m held
    void collect(int got) {
        for (;;) {
            if (got) {
                int w = 0, h = 0; jm_amddec_stream_info(&w, &h, dec);
                int cap = w * h * 3 / 2; if (cap < 16) cap = 16;
                tmp.resize((size_t)cap);
                int n = cap;
                if (jm_amddec_output_frame(tmp.data(), &n, dec) > 0) {
                    if (cb) cb(tmp.data(), n, user);
                    else fifo.emplace_back(tmp.begin(), tmp.begin() + n);
                }
            }
            if (fifo.size() >= kFifoHigh && !cb) break;
            got = 0;
            if (jm_amddec_poll_frame(&got, dec) != 0 || !got) break;
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
    int got = 0;
    if (jm_amddec_decode_frame(in_buf, n, &got, C(h)->dec) != 0) return -1;
    C(h)->collect(got);
    return n;
}
JM_EXPORT int jm_amdintel_output_frame(unsigned char *out_buf, int *out_len, jm_amdintel_handle h) {
    if (!h || !out_len) return -1;
    Ctx *c = C(h);
    std::lock_guard<std::mutex> lk(c->m);
    if (c->inited && c->fifo.empty()) { int got = 0; if (c->eof) jm_amddec_decode_frame(nullptr, 0, &got, c->dec); else jm_amddec_poll_frame(&got, c->dec);
        c->collect(got); }
    if (c->fifo.empty() || c->cb) { *out_len = 0; return -1; }
    std::vector<unsigned char> &f = c->fifo.front();
    if (!out_buf) { *out_len = (int)f.size(); return 0; }                 // size query (jm_intel_dec.h:74)
    if (*out_len < (int)f.size()) { *out_len = 0; return -2; }            // intel_dec.cpp:266-270
    memcpy(out_buf, f.data(), f.size()); *out_len = (int)f.size();
    c->fifo.pop_front();
    return 0;
}
JM_EXPORT int jm_amdintel_set_eof(int is_eof, jm_amdintel_handle h) {
    if (!h || !is_eof || !C(h)->inited) return h ? 0 : -1;
    std::lock_guard<std::mutex> lk(C(h)->m);
    if (!C(h)->eof) { C(h)->eof = true; int got = 0; jm_amddec_decode_frame(nullptr, 0, &got, C(h)->dec); C(h)->collect(got); }
    return 0;
}
JM_EXPORT char *jm_amdintel_info(jm_amdintel_handle h) { return h ? jm_amddec_show_dec_info(C(h)->dec) : nullptr; }
JM_EXPORT int jm_amdintel_get_stream_info(int *w, int *hh, float *fps, jm_amdintel_handle h) {
    if (!h || !w || !hh) return -1;
    if (fps) *fps = 0.0f;                                                // frame rate lives in the VUI; not tracked
    return jm_amddec_stream_info(w, hh, C(h)->dec);
}
JM_EXPORT int jm_amdintel_need_more_data(jm_amdintel_handle h) {
    if (!h) return 0;
    std::lock_guard<std::mutex> lk(C(h)->m);
    return !C(h)->eof && (C(h)->cb || C(h)->fifo.size() < kFifoHigh);
}
JM_EXPORT int jm_amdintel_free_buf_len(jm_amdintel_handle h) { return h ? kInputChunk : 0; }
JM_EXPORT int jm_amdintel_is_exit(jm_amdintel_handle h) {
    if (!h) return 1;
    Ctx *c = C(h);
    std::lock_guard<std::mutex> lk(c->m);
    if (!c->eof) return 0;
    if (c->fifo.empty() && !jm_amddec_is_exit(c->dec)) { int got = 0; jm_amddec_decode_frame(nullptr, 0, &got, c->dec); c->collect(got); }
    return c->fifo.empty() && jm_amddec_is_exit(c->dec);
}
JM_EXPORT int jm_amdintel_is_hw_support(void) { return jm_amddec_is_hw_support(); }
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
