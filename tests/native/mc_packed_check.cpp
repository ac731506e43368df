// tests/native/mc_packed_check.cpp -- host build of jmcodec_amd/csrc/mc_packed.h (the packed sample arithmetic of the inter reconstruction) behind a C
// ABI, so that tests/test_mc_packed.py can check it against a literal restatement of 8.4.2.2.1 / 8.4.2.2.2 without a GPU.  Test infrastructure only.
#include "../../jmcodec_amd/csrc/mc_packed.h"
using namespace jmamd::pk;
extern "C" {
uint32_t mcp_luma4(const uint32_t *win, int stride, int wr, int cb, int fx, int fy) { return mc_luma4(win, stride, wr, cb, fx, fy); }
uint32_t mcp_chroma_uv(uint32_t wa, uint32_t wb, int fx, int fy) { return mc_chroma_uv(wa, wb, chroma_weights(fx, fy)); }
uint32_t mcp_add_residual4(uint32_t pred, int r0, int r1, int r2, int r3) {
    return add_residual4(pred, ((uint32_t)r0 & 0xffffu) | ((uint32_t)r1 << 16), ((uint32_t)r2 & 0xffffu) | ((uint32_t)r3 << 16));
}
uint32_t mcp_add_residual_uv(uint32_t uv, int ru, int rv) { return add_residual_uv(uv, ru, rv); }
uint32_t mcp_lerp(uint32_t a, uint32_t b) { return lerp(a, b, kOnes); }
}
