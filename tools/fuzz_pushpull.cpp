// tools/fuzz_pushpull.cpp -- developer tool: the push / pull facade (jm_intel_dec_api.cpp: caller thread + feeder thread + the decoder's parse workers) driven
// like test_intel_dec.cpp:64-102 in parse-only mode (no GPU work), for ThreadSanitizer / AddressSanitizer builds of the host code:
//   make -C tools fuzz_pushpull_tsan && tools/_build/fuzz_pushpull_tsan stream.h264 [seed] [trials] [codec_type]
// Every trial: random push sizes, random fetch pattern (output_frame / callback), sometimes an early deinit with input still buffered, sometimes input after
// end of stream; the frame count of a complete run must equal the first trial's.
#include "../include/jm_amd_dec.h"
#include "../include/jm_amd_intel_dec.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <atomic>

static std::vector<unsigned char> read_all(const char *p) {
    std::vector<unsigned char> v; FILE *f = fopen(p, "rb"); if (!f) return v;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, f) != (size_t)n) v.clear(); fclose(f); return v;
}
static std::atomic<long> g_cb_frames{0};
static int on_frame(unsigned char *, int, void *) { g_cb_frames++; return 0; }

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s stream [seed] [trials] [codec_type]\n", argv[0]); return 2; }
    std::vector<unsigned char> data = read_all(argv[1]);
    if (data.size() < 200) { fprintf(stderr, "%s: missing or too short\n", argv[1]); return 2; }
    unsigned long long s = argc > 2 ? strtoull(argv[2], nullptr, 0) : 1; const int trials = argc > 3 ? atoi(argv[3]) : 50, codec = argc > 4 ? atoi(argv[4]) : 0;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(s >> 33); };
    std::vector<unsigned char> out(4 << 20);
    long want = -1, total = 0;
    for (int t = 0; t < trials; t++) {
        jm_amdintel_handle h = jm_amdintel_create_handle();
        jm_amddec_set_option(jm_amdintel_decoder(h), "parse_only", 1);
        const bool use_cb = t % 3 == 1, abandon = t % 7 == 6;
        if (use_cb) { g_cb_frames = 0; jm_amdintel_set_yuv_callback(nullptr, on_frame, h); }
        if (jm_amdintel_init(codec, 1, h) != 0) { fprintf(stderr, "init failed\n"); return 1; }
        size_t pos = 0; long frames = 0; bool eof = false; long guard = 0;
        while (!jm_amdintel_is_exit(h)) {
            if (++guard > 50000000) { fprintf(stderr, "trial %d: no end (pos %zu of %zu, frames %ld)\n", t, pos, data.size(), frames); return 1; }
            if (!eof && jm_amdintel_need_more_data(h)) {
                if (pos < data.size()) {
                    size_t n = 1 + rnd() % (t % 2 ? 70000u : 3000u); const size_t room = (size_t)jm_amdintel_free_buf_len(h);
                    if (n > room) n = room; if (n > data.size() - pos) n = data.size() - pos;
                    if (n) { const int r = jm_amdintel_input_data(data.data() + pos, (int)n, h); if (r < 0) { fprintf(stderr, "trial %d: input_data %d\n", t, r); return 1; } pos += (size_t)r; }
                } else { jm_amdintel_set_eof(1, h); eof = true;
                    if (t % 5 == 0 && jm_amdintel_input_data(data.data(), 100, h) >= 0) { fprintf(stderr, "trial %d: input after end of stream accepted\n", t); return 1; } }
            }
            if (abandon && pos > data.size() / 2) break;                          // deinit with input buffered, frames unfetched, the feeder mid-flight
            if (rnd() % 4) { int len = (int)out.size(); const int r = jm_amdintel_output_frame(out.data(), &len, h); if (r == 0 && !use_cb) frames++; }
        }
        if (use_cb) frames = g_cb_frames;
        if (t % 4 == 0) { int w = 0, hh = 0; float fr = 0; jm_amdintel_get_stream_info(&w, &hh, &fr, h); (void)jm_amdintel_info(h); }
        jm_amdintel_deinit(h);
        if (!abandon) { if (want < 0) want = frames; else if (frames != want) { fprintf(stderr, "trial %d: %ld frames, trial 0 had %ld\n", t, frames, want); return 1; } }
        total += frames;
    }
    printf("ok: %d trials, %ld frames (%ld per complete run)\n", trials, total, want);
    return 0;
}

#include "fuzz_stubs.h"
