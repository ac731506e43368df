// jmcodec_amd/csrc/chain_order.h -- the order of reconstruction groups INSIDE one key bucket of a chain launch's work list (engine.cpp, Engine::launch).
//
// Groups of one key are independent of each other (the key rule of chain_common.h orders buckets, not their contents), so the order inside a bucket is free --
// and decides on which XCD a group runs: work-list entry g becomes workgroups 2g and 2g + 1, which the dispatcher hands to XCDs (2g) % 8 and (2g + 1) % 8, so
// entry g lands on XCD pair g % 4.  Each XCD has its own L2; in plain insertion order the groups that read one stretch of a reference picture (the same
// 8-macroblock column in neighbouring rows: their windows overlap by 5 of 21 rows) were spread over all eight.  Here column c of picture i goes to XCD pair
// (c + i) % 4 whenever the bucket still holds such a group for the position at hand, else whatever class it holds most of (no padding entries).
// Host code, header-only; tests/test_chain_order.py checks that the result is a permutation of the bucket and that a position gets its class whenever one is left.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace jmamd {

// entry = picture << 16 | row * 32 + column (engine.cpp); its class: (column + picture) % 4
inline int chain_group_class(uint32_t e) { return (int)((e & 31u) + (e >> 16)) & 3; }

// appends bk[0 .. n) to out[n_out ..), n_out = the entry's position in the work list; tmp: scratch, grown as needed
inline void append_bucket_by_xcd(const uint32_t *bk, size_t n, uint32_t *out, int &n_out, std::vector<uint32_t> &tmp) {
    if (n < 2) { for (size_t i = 0; i < n; i++) out[n_out++] = bk[i]; return; }
    if (tmp.size() < 4 * n) tmp.resize(4 * n);
    uint32_t *q[4]; int qn[4] = {0, 0, 0, 0}, qi[4] = {0, 0, 0, 0};
    for (int j = 0; j < 4; j++) q[j] = tmp.data() + (size_t)j * n;
    for (size_t i = 0; i < n; i++) { const int j = chain_group_class(bk[i]); q[j][qn[j]++] = bk[i]; }
    for (size_t left = n; left; left--) {
        int j = n_out & 3;
        if (qi[j] == qn[j]) { int best = 0; for (int t = 1; t < 4; t++) if (qn[t] - qi[t] > qn[best] - qi[best]) best = t; j = best; }
        out[n_out++] = q[j][qi[j]++];
    }
}

}  // namespace jmamd
