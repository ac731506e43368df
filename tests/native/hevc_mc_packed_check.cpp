// tests/native/hevc_mc_packed_check.cpp -- host build of jmcodec_amd/csrc/hevc_mc_packed.h: the two passes of k_hevc_mc for one block, run lane by lane as
// the wave runs them, so that tests/test_hevc_mc_packed.py can compare them with a literal restatement of H.265 8.5.3.3.3 without a GPU.
#include "../../jmcodec_amd/csrc/hevc_mc_packed.h"
#include <cstring>
#include <vector>
using namespace jmamd;

extern "C" {
// plane: samples (luma) or interleaved Cb Cr bytes (chroma), `pitch` bytes per row; (x0b, y0): byte column / row of the window's first sample -- the caller
// keeps the window inside the plane (the device's border path only differs in how the tile is filled).  bw / bh: block size in samples (chroma: in pairs).
// out14: bh rows of bw (chroma: 2 * bw, Cb Cr interleaved) 14-bit intermediates.
void hmcp_block(const uint8_t *plane, int pitch, int x0b, int y0, int bw, int bh, int xf, int yf, int chroma, int *out14) {
    const int th = bh + (chroma ? 3 : 7), row_bytes = chroma ? 2 * (bw + 3) : bw + 7, qw = chroma ? bw >> 1 : bw >> 2;
    std::vector<uint32_t> tile(hpk::kMcTileDw, 0xdeadbeefu), hcol(16 * hpk::kMcColDw, 0xdeadbeefu);
    const int sh = x0b & 3, ndw = (sh + row_bytes + 3) >> 2;
    for (int r = 0; r < th; r++) for (int d = 0; d < ndw; d++) {
        uint32_t w; memcpy(&w, plane + (size_t)(y0 + r) * pitch + (x0b & ~3) + 4 * d, 4);
        tile[r * hpk::kMcRowDw + d] = w ^ 0x80808080u;
    }
    uint32_t ta = 0, tb = 0;
    if (chroma) ta = hpk::chroma_taps_h(xf); else hpk::luma_taps_h(xf, ta, tb);
    for (int lane = 0; lane < 64; lane++) { if (chroma) hpk::mc_pass1<true>(tile.data(), sh, qw, th, lane, ta, tb, hcol.data());
        else hpk::mc_pass1<false>(tile.data(), sh, qw, th, lane, ta, tb, hcol.data()); }
    for (int lane = 0; lane < 64; lane++) {
        const int row = lane / qw, q = lane - row * qw;
        if (lane >= bh * qw) continue;
        uint32_t tp[5] = {0, 0, 0, 0, 0};
        int o[4];
        if (chroma) { hpk::chroma_taps_v(yf, row & 1, tp); hpk::mc_pass2<true>(hcol.data(), row, q, tp, xf != 0, yf != 0, o); }
        else { hpk::luma_taps_v(yf, row & 1, tp); hpk::mc_pass2<false>(hcol.data(), row, q, tp, xf != 0, yf != 0, o); }
        for (int i = 0; i < 4; i++) out14[row * 4 * qw + 4 * q + i] = o[i];
    }
}
uint32_t hmcp_weigh_default4(const int *a, const int *b, int both) { return hpk::weigh_default4(a, b, both != 0); }
int hmcp_luma_tap(int f, int i) { return hpk::luma_tap(f, i); }
int hmcp_chroma_tap(int f, int i) { return hpk::chroma_tap(f, i); }
}
