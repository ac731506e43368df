// tests/native/hevc_resid_packed_check.cpp -- host build of jmcodec_amd/csrc/hevc_resid_packed.h: the inverse transform of one block as a wave of k_hevc_resid
// runs it (scatter into row pairs, first stage task by task, second stage task by task), for tests/test_hevc_resid_packed.py.
#include "../../jmcodec_amd/csrc/hevc_resid_packed.h"
#include "../../jmcodec_amd/csrc/hevc_tables.h"
#include <cstring>
#include <vector>
using namespace jmamd;

extern "C" {
// coefs: count entries position | value << 16 (position = row * n + column); res: n * n residuals, row-major.  dst != 0: the 4x4 DST.
void hrpc_block(const uint32_t *coefs, int count, int log2, int dst, int *res) {
    static uint32_t mp[hrp::kPairDw]; static bool built = false;
    if (!built) { hrp::build_pair_table(hevc_trans, hevc_dst, mp); built = true; }
    const int n = 1 << log2, nn = n * n;
    std::vector<uint32_t> dp(512, 0xdeadbeefu), g(512, 0xdeadbeefu);      // (garbage where the kernel's LDS holds leftovers of the previous block)
    int mj = 0, mx = 0;
    for (int k = 0; k < count; k++) { const int pos = coefs[k] & 1023; mj = mj > (pos >> log2) ? mj : pos >> log2; mx = mx > (pos & (n - 1)) ? mx : pos & (n - 1); }
    const int jpmax = mj >> 1, cwp = hrp::pad_cols(mx + 1), cw = cwp < n ? cwp : n, lcw = hrp::log2_of(cw);
    for (int k = 0; k < ((jpmax + 1) << log2); k++) dp[k] = 0;
    int16_t *d16 = (int16_t *)dp.data();
    for (int k = 0; k < count; k++) { const int pos = coefs[k] & 1023; d16[hrp::pair_slot(pos >> log2, pos & (n - 1), log2)] = (int16_t)(coefs[k] >> 16); }
    const uint32_t *mp_n = mp + hrp::pair_off(log2, dst != 0);
    int16_t *g16 = (int16_t *)g.data();
    for (int t = 0; t < (n << lcw); t++) { int gi; const int v = hrp::col_task(t, log2, lcw, jpmax, mp_n, dp.data(), gi); g16[gi] = (int16_t)v; }
    for (int t = 0; t < nn / 2; t++) dp[t] = hrp::row_task(t, log2, cw >> 1, mp_n, g.data());
    const int16_t *r16 = (const int16_t *)dp.data();
    for (int k = 0; k < nn; k++) res[k] = r16[k];
}
int hrpc_trans(int j, int y) { return hevc_trans[j][y]; }
int hrpc_dst(int j, int y) { return hevc_dst[j][y]; }
}
