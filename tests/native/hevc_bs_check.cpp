// tests/native/hevc_bs_check.cpp -- host build of jmcodec_amd/csrc/hevc_bs.h: the maps painted from job lists and the strength of every edge segment, exactly as
// k_hevc_bs_clear / k_hevc_bs_raster / k_hevc_bs run them, for tests/test_hevc_bs.py.
#include "../../jmcodec_amd/csrc/hevc_bs.h"
#include <cstring>
#include <vector>
using namespace jmamd;

extern "C" {
// pus: n_pus records of 10 ints (x y w h slot0 slot1 mv0x mv0y mv1x mv1y); itbs / tbs: records of 3 ints (x y log2) -- the LUMA intra blocks and the luma
// transform blocks with cbf_luma = 1; db_flags: one byte per CTB; qp8: one byte per 8x8.  bs_v: (w / 8) * (h / 4) bytes, bs_h: (w / 4) * (h / 8).
void hbs_run(int w, int h, int ctb_log2, const int *pus, int n_pus, const int *itbs, int n_itbs, const int *tbs, int n_tbs, const uint8_t *db_flags,
    const uint8_t *qp8, uint8_t *bs_v, uint8_t *bs_h) {
    const int w4 = w >> 2, h4 = h >> 2, cells = w4 * h4, ctb_w = (w + (1 << ctb_log2) - 1) >> ctb_log2, ctb_h = (h + (1 << ctb_log2) - 1) >> ctb_log2;
    std::vector<uint32_t> pu_map(cells, 0xdeadbeefu);
    std::vector<uint8_t> flags(4 * cells, 0);                        // (k_hevc_bs_clear)
    std::vector<HevcPu> P(n_pus > 0 ? n_pus : 1);
    for (int i = 0; i < n_pus; i++) { const int *r = pus + 10 * i; HevcPu &p = P[i]; memset(&p, 0, sizeof p);
        p.x = (uint16_t)r[0]; p.y = (uint16_t)r[1]; p.w = (uint8_t)r[2]; p.h = (uint8_t)r[3]; p.slot0 = (int8_t)r[4]; p.slot1 = (int8_t)r[5];
        p.mv0[0] = (int16_t)r[6]; p.mv0[1] = (int16_t)r[7]; p.mv1[0] = (int16_t)r[8]; p.mv1[1] = (int16_t)r[9]; }
    std::vector<HevcCtb> ctbs(ctb_w * ctb_h);
    for (int i = 0; i < ctb_w * ctb_h; i++) { memset(&ctbs[i], 0, sizeof(HevcCtb)); ctbs[i].db_flags = db_flags[i]; }
    const hbs::Maps m = hbs::maps_of(w, h, pu_map.data(), flags.data());
    for (int i = 0; i < n_pus; i++) hbs::paint_pu(m, i, P[i]);
    for (int i = 0; i < n_itbs; i++) hbs::mark(m, itbs[3 * i], itbs[3 * i + 1], 1 << itbs[3 * i + 2], m.f_intra);
    for (int i = 0; i < n_tbs; i++) hbs::mark(m, tbs[3 * i], tbs[3 * i + 1], 1 << tbs[3 * i + 2], m.f_cbf);
    const int w8 = w >> 3, h8 = h >> 3;
    for (int idx = 0; idx < w8 * h4; idx++) bs_v[idx] = (uint8_t)hbs::edge_strength(m, 0, (idx % w8) * 8, (idx / w8) * 4, ctb_log2, ctb_w, ctbs.data(), P.data(), qp8, w8);
    for (int idx = 0; idx < w4 * h8; idx++) bs_h[idx] = (uint8_t)hbs::edge_strength(m, 1, (idx % w4) * 4, (idx / w4) * 8, ctb_log2, ctb_w, ctbs.data(), P.data(), qp8, w8);
}
}
