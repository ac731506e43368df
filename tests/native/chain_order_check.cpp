// tests/native/chain_order_check.cpp -- jmcodec_amd/csrc/chain_order.h behind a C ABI for tests/test_chain_order.py.  Test infrastructure only.
#include "../../jmcodec_amd/csrc/chain_order.h"
extern "C" {
// appends the bucket to out at position *pos (updated); returns the new position
int co_append(const uint32_t *bk, int n, uint32_t *out, int pos) {
    static std::vector<uint32_t> tmp;
    jmamd::append_bucket_by_xcd(bk, (size_t)n, out, pos, tmp);
    return pos;
}
int co_class(uint32_t e) { return jmamd::chain_group_class(e); }
}
