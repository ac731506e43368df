// tests/native/intra8_packed_check.cpp -- host build of jmcodec_amd/csrc/intra8_packed.h (Intra8x8 prediction on packed bytes) behind a C ABI, so that
// tests/test_intra8_packed.py can check it against a literal restatement of H.264 8.3.2.2 without a GPU.  Test infrastructure only.
#include "../../jmcodec_amd/csrc/intra8_packed.h"
using namespace jmamd::pk;
extern "C" {
// left[8] = p[-1, 0..7], corner = p[-1, -1], top[16] = p[0..15, -1]; junk: what the other bytes of the dwords a lane loads hold (must not matter).
// out[64] = the predicted block, raster order.
void i8p_block(const uint8_t *left, int corner, const uint8_t *top, uint32_t junk, int a, int b, int c, int d, int mode, uint8_t *out) {
    I8Edge e;
    e.tl = (junk & 0x00ffffffu) | (uint32_t)corner << 24;
    auto dw = [&](int i) { return (uint32_t)top[i] | (uint32_t)top[i + 1] << 8 | (uint32_t)top[i + 2] << 16 | (uint32_t)top[i + 3] << 24; };
    e.t0 = dw(0); e.t1 = dw(4); e.r0 = dw(8); e.r1 = dw(12);
    for (int i = 0; i < 8; i++) e.l[i] = ((junk * (i + 3)) & 0x00ffffffu) | (uint32_t)left[i] << 24;
    uint32_t F[7];
    i8_filtered_path(e, a != 0, b != 0, c != 0, d != 0, F);
    const int dc = i8_dc(F, a != 0, b != 0);
    for (int lane = 0; lane < 16; lane++) {
        const uint32_t p = i8_predict4(F, i8_sel_entry(mode, lane), dc);
        for (int j = 0; j < 4; j++) out[(lane >> 1) * 8 + (lane & 1) * 4 + j] = (uint8_t)(p >> (8 * j));
    }
}
// the filtered path alone: F[25]
void i8p_path(const uint8_t *left, int corner, const uint8_t *top, int a, int b, int c, int d, uint8_t *out) {
    I8Edge e;
    e.tl = (uint32_t)corner << 24;
    auto dw = [&](int i) { return (uint32_t)top[i] | (uint32_t)top[i + 1] << 8 | (uint32_t)top[i + 2] << 16 | (uint32_t)top[i + 3] << 24; };
    e.t0 = dw(0); e.t1 = dw(4); e.r0 = dw(8); e.r1 = dw(12);
    for (int i = 0; i < 8; i++) e.l[i] = (uint32_t)left[i] << 24;
    uint32_t F[7];
    i8_filtered_path(e, a != 0, b != 0, c != 0, d != 0, F);
    for (int k = 0; k < 25; k++) out[k] = (uint8_t)(F[k >> 2] >> (8 * (k & 3)));
}
}
