// jmcodec_amd/csrc/jm_intel_dec_api.cpp -- push/pull decode API of /root/reference/intel_dec/jm_intel_dec.h over the MI355X engine
// (BASELINE config 0's call shape, test_intel_dec/test_intel_dec.cpp:64-102).  The reference runs a worker thread that pulls from an
// input bitstream buffer and queues decoded surfaces (intel_dec.cpp:46-81, 189-332); here the engine already is asynchronous and the decoder already
// keeps its display frames in device memory in display order, so the facade only adapts the call protocol and never holds a frame itself:
//   input_data   = a copy into the input buffer; a feeder thread hands it to the decoder (jm_amddec_push_data: a frame the caller has not fetched stays current);
//   output_frame = take the next finished display frame and let the decoder put it into the CALLER'S buffer with its one copy-engine transfer --
//                  one copy per frame, the same as jm_nvdec_output_frame (round 5 had three: device -> a scratch buffer -> a fresh vector -> the caller);
//   need_more_data / free_buf_len describe a 1 MB input buffer, as in the reference (intel_dec.cpp:343-360): input is wanted while more than half of it is free.
#include "../../include/jm_amd_dec.h"
#include "../../include/jm_amd_intel_dec.h"
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include <cstring>

#define JM_EXPORT __attribute__((visibility("default")))

namespace {
constexpr int kInputChunk = 1 << 20;      // capacity of the input buffer = what free_buf_len reports when it is empty (the reference's starts at 1 MB, intel_dec.h)
constexpr long long kFramesHigh = 64;     // display frames waiting for the caller: more than that and the feeder holds its next piece back
constexpr int kPiece = 64 << 10;          // the feeder hands the decoder pieces of this size (it may block in the decoder: every picture needs a job slot)
constexpr int kWaitUs = 2000;             // longest sleep of output_frame for a frame that is being decoded, while no input is wanted
// The reference decodes on a worker thread that pulls from the input bitstream buffer (intel_dec.cpp:46-81) while the application pushes and pulls; so here:
// input_data copies into the buffer and returns, the FEEDER thread hands the decoder the buffer piece by piece -- it, not the caller, waits when the
// decoder's job slots are all taken -- and the caller's thread only ever copies frames out.  (A first version of round 6 pushed on the caller's thread: a
// 1 MB push holds some 45 pictures of the 1080p C1 stream, the push waited for job slots, and while it waited nobody fetched frames: 0.80-0.83 of the
// NAL-per-call rate.)
struct Ctx {
    jm_amddec_handle dec = nullptr;
    std::mutex m;                         // the caller's side: `have`, the callback
    jm_amdintel_yuv_callback cb = nullptr; void *user = nullptr;
    std::vector<unsigned char> cb_buf;    // callback mode: the one buffer every frame is delivered in
    bool eof = false, inited = false;
    bool have = false;                    // the decoder holds a current frame that nobody fetched yet
    // the input buffer and its feeder
    std::mutex im; std::condition_variable icv;
    std::deque<std::vector<unsigned char>> inq; size_t buffered = 0;
    bool in_eof = false, stop = false;
    std::atomic<bool> flushed{false}, feed_failed{false};
    std::thread feeder;
    void feed_loop() {
        for (;;) {
            std::vector<unsigned char> chunk;
            {
                std::unique_lock<std::mutex> lk(im);
                icv.wait(lk, [&] { return stop || !inq.empty() || in_eof; });
                if (stop) return;
                if (inq.empty()) { lk.unlock(); if (jm_amddec_push_eos(dec) != 0) feed_failed = true; flushed = true; return; }
                chunk = std::move(inq.front()); inq.pop_front();
            }
            for (size_t o = 0; o < chunk.size() && !feed_failed;) {
                { std::lock_guard<std::mutex> lk(im); if (stop) return; }          // (deinit in the middle of a chunk)
                // (frames nobody fetches hold output slots: wait for the caller rather than decode the whole stream into memory)
                while (!cb && jm_amddec_get_stat(dec, "frames_waiting") >= kFramesHigh) {
                    std::unique_lock<std::mutex> lk(im);
                    if (icv.wait_for(lk, std::chrono::microseconds(500), [&] { return stop; })) return;
                }
                const size_t n = chunk.size() - o < (size_t)kPiece ? chunk.size() - o : (size_t)kPiece;
                if (jm_amddec_push_data(chunk.data() + o, (int)n, dec) != 0) feed_failed = true;
                o += n;
                { std::lock_guard<std::mutex> lk(im); buffered -= n; }
            }
        }
    }
    // make the next finished display frame current (m held).  After the end of the stream went in, the decoder drains: it blocks until the next frame is there
    // or none is left.
    bool take(int wait_us) {
        if (have || !inited) return have;
        int got = 0;
        if (flushed) jm_amddec_decode_frame(nullptr, 0, &got, dec);
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
    size_t room() { std::lock_guard<std::mutex> lk(im); return buffered < (size_t)kInputChunk ? (size_t)kInputChunk - buffered : 0; }
};
inline Ctx *C(void *h) { return static_cast<Ctx *>(h); }
}  // namespace

extern "C" {
JM_EXPORT jm_amdintel_handle jm_amdintel_create_handle(void) { Ctx *c = new Ctx(); c->dec = jm_amddec_create_handle(); return c; }
JM_EXPORT int jm_amdintel_init(int codec_type, int out_fmt, jm_amdintel_handle h) {
    if (!h || C(h)->inited) return -1;
    int rc = jm_amddec_init(codec_type, out_fmt, nullptr, 0, C(h)->dec);
    C(h)->inited = rc == 0;
    if (rc == 0) C(h)->feeder = std::thread([c = C(h)] { c->feed_loop(); });
    return rc;
}
JM_EXPORT int jm_amdintel_deinit(jm_amdintel_handle h) {
    if (!h) return -1;
    Ctx *c = C(h);
    { std::lock_guard<std::mutex> lk(c->im); c->stop = true; }
    c->icv.notify_all();
    if (c->feeder.joinable()) c->feeder.join();
    jm_amddec_deinit(c->dec); delete c; return 0;
}
JM_EXPORT int jm_amdintel_set_yuv_callback(void *user, jm_amdintel_yuv_callback cb, jm_amdintel_handle h) {
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(C(h)->m); C(h)->cb = cb; C(h)->user = user; return 0;
}
// intel_dec_put_input_data (intel_dec.cpp:189-234): copy into the input buffer (which grows when the caller pushes more than free_buf_len reported)
JM_EXPORT int jm_amdintel_input_data(unsigned char *in_buf, int n, jm_amdintel_handle h) {
    if (!h || !C(h)->inited || !in_buf || n <= 0 || C(h)->eof || C(h)->feed_failed) return -1;
    Ctx *c = C(h);
    { std::lock_guard<std::mutex> lk(c->im); c->inq.emplace_back(in_buf, in_buf + n); c->buffered += (size_t)n; }
    c->icv.notify_all();
    std::lock_guard<std::mutex> lk(c->m);
    c->deliver();
    return n;
}
JM_EXPORT int jm_amdintel_output_frame(unsigned char *out_buf, int *out_len, jm_amdintel_handle h) {
    if (!h || !out_len) return -1;
    Ctx *c = C(h);
    std::lock_guard<std::mutex> lk(c->m);
    if (c->cb) { c->deliver(); *out_len = 0; return -1; }                 // frames go to the callback
    // (no frame ready: sleep for it only when the caller has nothing better to do -- no input is wanted)
    if (!c->take(c->room() < (size_t)kInputChunk / 2 || c->eof ? kWaitUs : 0)) { *out_len = 0; return -1; }   // intel_dec.cpp:251-255
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
    Ctx *c = C(h);
    { std::lock_guard<std::mutex> lk(c->m); c->eof = true; }
    { std::lock_guard<std::mutex> lk(c->im); c->in_eof = true; }          // the feeder sends the end of the stream behind what is still buffered
    c->icv.notify_all();
    std::lock_guard<std::mutex> lk(c->m);
    c->deliver();
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
// intel_dec_need_more_data (intel_dec.cpp:343-352): the input buffer has room for more than half of its capacity
JM_EXPORT int jm_amdintel_need_more_data(jm_amdintel_handle h) {
    if (!h) return 0;
    Ctx *c = C(h);
    { std::lock_guard<std::mutex> lk(c->m); if (c->eof) return 0; }
    return c->room() > (size_t)kInputChunk / 2;
}
// intel_dec_get_input_free_buf_len (intel_dec.cpp:357-360)
JM_EXPORT int jm_amdintel_free_buf_len(jm_amdintel_handle h) { return h ? (int)C(h)->room() : 0; }
JM_EXPORT int jm_amdintel_is_exit(jm_amdintel_handle h) {
    if (!h) return 1;
    Ctx *c = C(h);
    std::lock_guard<std::mutex> lk(c->m);
    if (!c->eof) return 0;
    if (c->cb) c->deliver();
    else if (!c->have && c->flushed && !jm_amddec_is_exit(c->dec)) c->take(0);   // the drain call that finds the queue empty is what ends the decoder (nv_dec.cpp:460-466)
    return !c->have && c->flushed && jm_amddec_is_exit(c->dec);
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
