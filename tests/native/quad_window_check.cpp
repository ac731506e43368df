// tests/native/quad_window_check.cpp -- host build of the shared-window geometry of chain launches (jmcodec_amd/csrc/mc_packed.h: quad_geometry, chroma_pairs;
// the loops below restate the cooperative load of recon_device.h index for index) behind a C ABI: tests/test_quad_window.py checks that four macroblocks
// predicted out of ONE shared window get exactly the samples they get out of their private windows, and that no byte outside the loaded extent is ever used
// (the rest of the window is filled with noise that differs between two runs).  Test infrastructure only.
#include "../../jmcodec_amd/csrc/mc_packed.h"
#include <stddef.h>
#include <vector>
using namespace jmamd::pk;
extern "C" {
// pic: luma plane (pitch bytes per row) followed by the interleaved chroma plane at chroma_offset; votes: 4 x {slot, xi, yi, 2 * cxi, cyi}; frac: 4 x {fx, fy,
// cfx, cfy}.  out_shared / out_private: 4 x (256 luma + 128 chroma) bytes.  Returns 1 when the geometry accepts the four, 0 when not (nothing written).
int qw_check(const uint8_t *pic, int pitch, int chroma_offset, const int *votes, const int *frac, uint32_t noise, uint8_t *out_shared, uint8_t *out_private) {
    int vote[4][8];
    for (int w = 0; w < 4; w++) for (int k = 0; k < 5; k++) vote[w][k] = votes[w * 5 + k];
    const QuadGeom g = quad_geometry(vote);
    if (!g.ok) return 0;
    auto ld = [&](const uint8_t *plane, int off) {
        return (uint32_t)plane[off] | (uint32_t)plane[off + 1] << 8 | (uint32_t)plane[off + 2] << 16 | (uint32_t)plane[off + 3] << 24; };
    std::vector<uint32_t> qwin(kQuadRows * kQuadStride), qcwin(kQuadChromaRows * kQuadStride);
    for (size_t i = 0; i < qwin.size(); i++) qwin[i] = noise * (uint32_t)(2654435761u + i * 40503u);
    for (size_t i = 0; i < qcwin.size(); i++) qcwin[i] = noise * (uint32_t)(97u + i * 7919u);
    const uint8_t *rc = pic + chroma_offset;
    for (int wave = 0; wave < 4; wave++) for (int lane = 0; lane < 64; lane++) {       // the cooperative load, as recon_device.h has it
        const int sub = wave * 2 + (lane >> 5), dwi = lane & 31;
        for (int k = 0; k < 4; k++) {
            const int rr = sub + 8 * k;
            const uint32_t v = ld(pic, (g.y0 + (rr < g.nrow - 1 ? rr : g.nrow - 1)) * pitch + g.x0 + (dwi < g.ndw - 1 ? dwi : g.ndw - 1) * 4);
            if (rr < g.nrow && dwi < g.ndw) qwin[rr * kQuadStride + dwi] = v ^ kSign;
        }
        for (int k = 0; k < 2; k++) {
            const int rr = sub + 8 * k;
            const uint32_t v = ld(rc, (g.cy0 + (rr < g.ncrow - 1 ? rr : g.ncrow - 1)) * pitch + g.cx0 + (dwi < g.ncdw - 1 ? dwi : g.ncdw - 1) * 4);
            if (rr < g.ncrow && dwi < g.ncdw) qcwin[rr * kQuadStride + dwi] = v;
        }
    }
    for (int w = 0; w < 4; w++) {
        const int xi = vote[w][1], yi = vote[w][2], cx2 = vote[w][3], cyi = vote[w][4];
        const int fx = frac[w * 4], fy = frac[w * 4 + 1], cfx = frac[w * 4 + 2], cfy = frac[w * 4 + 3];
        // the private windows of the one-window path: 21 rows x 6 dwords from xi & ~3, 9 rows x 5 dwords from cx2 & ~3 (the rest: noise)
        uint32_t w16[21 * 6 + 8], cw[9 * 5 + 4];
        for (int i = 0; i < 21 * 6 + 8; i++) w16[i] = noise * (uint32_t)(31u + i);
        for (int i = 0; i < 9 * 5 + 4; i++) cw[i] = noise * (uint32_t)(77u + i);
        for (int i = 0; i < 126; i++) w16[i] = ld(pic, (yi + i / 6) * pitch + (xi & ~3) + (i % 6) * 4) ^ kSign;
        for (int i = 0; i < 45; i++) cw[i] = ld(rc, (cyi + i / 5) * pitch + (cx2 & ~3) + (i % 5) * 4);
        uint8_t *os = out_shared + w * 384, *op = out_private + w * 384;
        for (int lane = 0; lane < 64; lane++) {
            const int py = lane >> 2, px = (lane & 3) * 4;
            const uint32_t a = mc_luma4(qwin.data(), kQuadStride, py + (yi - g.y0), px + (xi - g.x0), fx, fy), b = mc_luma4(w16, 6, py, px + (xi & 3), fx, fy);
            for (int j = 0; j < 4; j++) { os[py * 16 + px + j] = (uint8_t)(a >> (8 * j)); op[py * 16 + px + j] = (uint8_t)(b >> (8 * j)); }
            const int cx = lane & 7, cy = lane >> 3;
            uint32_t wa, wb, va, vb;
            chroma_pairs(qcwin.data() + (cyi - g.cy0) * kQuadStride, kQuadStride, cy, (cx2 - g.cx0) + 2 * cx, wa, wb);
            chroma_pairs(cw, 5, cy, (cx2 & 3) + 2 * cx, va, vb);
            const uint32_t ua = mc_chroma_uv(wa, wb, chroma_weights(cfx, cfy)), ub = mc_chroma_uv(va, vb, chroma_weights(cfx, cfy));
            os[256 + (cy * 8 + cx) * 2] = (uint8_t)ua; os[256 + (cy * 8 + cx) * 2 + 1] = (uint8_t)(ua >> 8);
            op[256 + (cy * 8 + cx) * 2] = (uint8_t)ub; op[256 + (cy * 8 + cx) * 2 + 1] = (uint8_t)(ub >> 8);
        }
    }
    return 1;
}
}
