// tests/cabac_engine_check.cpp -- compiled and run by tests/test_host_parser.py::test_cabac_engine_shortcuts_equal_the_bin_by_bin_decoder.
// The arithmetic decoder's shortcuts (jmcodec_amd/csrc/h264_cabac.h) against the operations of 9.3.3.2.3 taken one bin at a time:
//   cabac_quotient   n bypass bins as a quotient by reciprocal multiplication  -- every range, x around every multiple of the range
//   bypass_bits / unary (Cabac and CabacRegs)                                  -- random and all-ones data, the engines must stay in step
#include "../jmcodec_amd/csrc/h264_cabac.h"
#include <cstdio>
#include <random>
using namespace jmamd;
int main() {
    for (uint32_t r = 256; r <= 511; r++) for (uint64_t k = 0; k <= 65536; k++) for (int d = -1; d <= 1; d++) {
        const uint64_t x = k * r + (uint64_t)(int64_t)d;
        if ((int64_t)x < 0 || x >= ((uint64_t)r << 16)) continue;
        if (cabac_quotient(x, r) != x / r) { printf("quotient: range %u x %llu\n", r, (unsigned long long)x); return 1; }
    }
    std::mt19937_64 g(5);
    for (int trial = 0; trial < 20000; trial++) {
        uint8_t buf[256];
        for (auto &b : buf) b = (uint8_t)(g() >> (trial % 3 == 0 ? 56 : 0) | (trial % 5 == 0 ? 0xf0 : 0));
        if (trial % 7 == 0) for (int i = 0; i < 256; i++) buf[i] |= 0xfe;          // long runs of 1 bins
        buf[0] &= 0x7f;                                                            // codIOffset < 510 (9.3.1.2)
        Cabac a, b; a.init_engine(buf, buf + 256); b.init_engine(buf, buf + 256);
        for (int i = 0; i < CABAC_N_CTX; i++) a.state[i] = b.state[i] = (Cabac::State)(g() % 126);
        for (int step = 0; step < 300; step++) {
            const int op = (int)(g() % 4);
            if (op == 0) { const int c = (int)(g() % 64); if (a.decision(c) != b.decision(c)) { printf("decision\n"); return 1; } }
            else if (op == 1) { const int lim = 1 + (int)(g() % 33); int q = 0; while (q < lim && a.bypass()) q++;
                const int u = b.unary(lim); if (q != u) { printf("unary: %d, bin by bin %d, limit %d\n", u, q, lim); return 1; } }
            else if (op == 2) { const int n = 1 + (int)(g() % 16); uint32_t q = 0; for (int i = 0; i < n; i++) q = q << 1 | (uint32_t)a.bypass();
                if (q != b.bypass_bits(n)) { printf("bypass_bits %d\n", n); return 1; } }
            else { CabacRegs r(b); const int lim = 1 + (int)(g() % 33); int q = 0; while (q < lim && a.bypass()) q++;
                const int u = r.unary(lim); r.commit(); if (q != u) { printf("CabacRegs::unary\n"); return 1; } }
            if (a.val != b.val || a.pos != b.pos || a.range != b.range || a.ptr != b.ptr) { printf("the engines diverged after operation %d\n", op); return 1; }
        }
    }
    printf("ok\n");
    return 0;
}
