// tools/fuzz_host.cpp -- developer tool: feeds (corrupted) Annex-B streams through the product's host pipeline in parse-only
// mode (no GPU work) so that it can be run under AddressSanitizer / UBSan on the CPU build:
//   make -C tools fuzz_host_asan && tools/_build/fuzz_host_asan stream.h264 [seed] [trials] [codec_type: 0 H.264, 1 HEVC]
#include "../include/jm_amd_dec.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static std::vector<unsigned char> read_all(const char *p) {
    std::vector<unsigned char> v; FILE *f = fopen(p, "rb"); if (!f) return v;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, f) != (size_t)n) v.clear(); fclose(f); return v;
}
int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s stream.h264|stream.h265 [seed] [trials] [codec_type]\n", argv[0]); return 2; }
    std::vector<unsigned char> base = read_all(argv[1]);
    if (base.size() < 200) { fprintf(stderr, "%s: missing or shorter than 200 bytes\n", argv[1]); return 2; }
    unsigned long long s = argc > 2 ? strtoull(argv[2], nullptr, 0) : 1; int trials = argc > 3 ? atoi(argv[3]) : 100;
    const int codec = argc > 4 ? atoi(argv[4]) : 0;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(s >> 33); };
    long frames = 0;
    for (int t = 0; t < trials; t++) {
        std::vector<unsigned char> b = base;
        if (t > 0 && t % 3 == 2) {
            // parameter-set trials: the leading SPS / PPS / first slice header (roughly the first 64 bytes) get the damage -- bit flips, and
            // runs of zero bits that turn a short ue(v) into a 32-bit-class value (the crafted-SPS cases of tests/test_host_parser.py)
            const size_t lim = b.size() < 64 ? b.size() : 64;
            for (int k = 0; k < 1 + t % 4; k++) { size_t p = 4 + rnd() % (lim - 4); if (rnd() % 3 == 0) { b[p] = 0; if (p + 1 < lim && rnd() % 2) b[p + 1] = 0;
                } else b[p] ^= (unsigned char)(1u << (rnd() % 8)); }
        } else if (t > 0) { for (int k = 0; k < 1 + t % 6; k++) { size_t p = 30 + rnd() % (b.size() - 30); b[p] ^= (unsigned char)(1u << (rnd() % 8)); }
            if (t % 4 == 0) b.resize(100 + rnd() % (b.size() - 100)); }
        void *h = jm_amddec_create_handle();
        jm_amddec_set_option(h, "parse_only", 1);
        if (t % 2) jm_amddec_set_option(h, "digest", 1);
        jm_amddec_init(codec, 1, nullptr, 0, h);
        size_t pos = 0; int got = 0;
        while (pos < b.size()) { size_t n = 1 + rnd() % 4096; if (n > b.size() - pos) n = b.size() - pos;
            jm_amddec_decode_frame(b.data() + pos, (int)n, &got, h); frames += got; pos += n; }
        for (int i = 0; i < 100000 && !jm_amddec_is_exit(h); i++) { jm_amddec_decode_frame(nullptr, 0, &got, h); frames += got; }
        jm_amddec_deinit(h);
    }
    printf("ok: %d trials, %ld frames\n", trials, frames);
    return 0;
}

#include "fuzz_stubs.h"
