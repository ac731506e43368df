#!/usr/bin/env python3
"""Brute-force check of the work-list rule of chain launches (jmcodec_amd/csrc/engine.cpp, Engine::launch): every dependency of a reconstruction group
must have a SMALLER key than the group, because the dispatcher starts workgroups in work-list order and a waiting workgroup keeps its slot -- then the
unfinished group with the smallest key is always resident and can run, whatever the occupancy.  (Round 4: two violations were found with this model
after chain launches of 4 / 8 streams had given up on the GPU: one slope for a whole launch, and no extra room behind a picture with the intra role.)

The model restates, macroblock by macroblock, what the device code waits for:
  * a reconstruction group of picture P+1 (8 macroblocks of a row) waits until the deblocking of P has stored the last sample its reference windows
    touch (`ChainView::wait_final`, chain_common.h): `fin` of one or two bands of P >= X + L*Y + 1 (+ 1: one-row schedule);
  * a deblocking band of P completes iteration s only after (deblock_device.h `deblock_band_body`) the reconstruction bits of every macroblock it
    prefetches up to step s + DEPTH, and after the band above has published step s + DEPTH + kAbove (published kPubLag steps late, every `pub` steps);
  * with the intra role (`k_chain_i`), the band is gated by P's intra wavefront instead (x + 2y steps, `intra_device.h`), which waits for the bits.
Keys and spacing are the engine's (`key()`, `spacing()` below mirror Engine::launch; tests/test_chain_keys.py checks that the constants agree).
The rule orders BUCKETS of equal key; inside a bucket the engine is free (round 5: groups dealt by XCD, jmcodec_amd/csrc/chain_order.h -- a permutation of the
bucket, tests/test_chain_order.py).

    python3 tools/chain_keys.py [mb_w mb_h]        # prints, per case, (largest needed key of P) - (key of the dependent group): must be < 0
"""
import re
import sys
from functools import lru_cache

BR = 16                                   # kBandRows
DEPTH, PUB, KPUBLAG = 2, 2, 3             # deblock_depth(), deblock_pub(), kPubLag
ROW_LAG = 1                               # kRowLag (one macroblock row per deblocking step)
K_BAND_LAG, KEY_SLACK, INTRA_EXTRA, CHAIN_LAG = 8, 2, 24, 24


def constants_in_sources(root):
    """The same constants as the sources state them (for the test)."""
    eng = open(f"{root}/jmcodec_amd/csrc/engine.cpp").read()
    engh = open(f"{root}/jmcodec_amd/csrc/engine.h").read()
    deb = open(f"{root}/jmcodec_amd/csrc/deblock_device.h").read()
    lds = open(f"{root}/jmcodec_amd/csrc/deblock_lds.hip").read()
    com = open(f"{root}/jmcodec_amd/csrc/chain_common.h").read()
    g = lambda pat, s: int(re.search(pat, s).group(1))
    return dict(BR=g(r"constexpr int kBandRows = (\d+);", deb), DEPTH=g(r"constexpr int kDeblockDepth = (\d+),", lds), PUB=g(r"kDeblockPub = (\d+);", lds),
                KPUBLAG=g(r"constexpr int kPubLag = (\d+);", deb), ROW_LAG=g(r"constexpr int kRowLag = (\d+);", com),
                K_BAND_LAG=g(r"constexpr int kBandLag = (\d+),", eng), KEY_SLACK=g(r"kKeySlack = (\d+),", eng), INTRA_EXTRA=g(r"kIntraExtra = (\d+);", eng),
                CHAIN_LAG=g(r"constexpr int kMinChainLag = (\d+);", engh))


def check(mb_w, mb_h, p_intra, g_intra, verbose=False):
    """Largest (needed key of P) - key(G) over all groups G of the picture after P."""
    L = ROW_LAG
    k_above = 2 - L
    store_lag = 1 if L == 1 else 0
    slope_p, slope_g = (2 if p_intra else L), (2 if g_intra else L)

    def rows_of(b):
        r0 = b * BR
        return r0, min(BR, mb_h - r0)

    def key_p(r, c):
        return slope_p * r + 8 * c + K_BAND_LAG * (r // BR)          # base(P) = 0

    @lru_cache(None)
    def intra_iter(b, s):                 # P's intra band b completes iteration s (x + 2y steps)
        r0, rows = rows_of(b)
        sb, se = 2 * r0, mb_w - 1 + 2 * (r0 + rows - 1)
        s = min(s, se)
        if s < sb:
            return -10 ** 9
        m = max([key_p(y, min(s + 1 - 2 * y, mb_w - 1) // 8) for y in range(r0, r0 + rows) if s + 1 - 2 * y >= 0] or [-10 ** 9])
        if b > 0:
            m = max(m, intra_iter(b - 1, s + 1))          # wait_above(s + 1)
        return m

    def ifin(b, v):                       # ifin[b] >= v: published as s - 1 after iteration s
        r0, rows = rows_of(b)
        return intra_iter(b, min(v + 1, mb_w - 1 + 2 * (r0 + rows - 1)))

    @lru_cache(None)
    def deb_iter(b, s):                   # P's deblocking band b completes iteration s
        r0, rows = rows_of(b)
        sb, se = L * r0, mb_w - 1 + L * (r0 + rows - 1) + store_lag
        s = min(s, se)
        if s < sb:
            return -10 ** 9
        m = -10 ** 9
        for y in range(r0, r0 + rows):
            x = min(s + DEPTH - L * y, mb_w - 1)
            if x < 0:
                continue
            if p_intra:                   # AFTER_INTRA: the intra wavefront has passed (x + 1, y + 1)
                m = max(m, ifin(y // BR, x + 2 * y + 4), ifin(min(y + 1, mb_h - 1) // BR, x + 2 * y + 4))
            else:
                m = max(m, key_p(y, x // 8))
        if b > 0:
            r0u, rowsu = rows_of(b - 1)
            sbu, seu = L * r0u, mb_w - 1 + L * (r0u + rowsu - 1) + store_lag
            want = s + DEPTH + k_above
            if want > seu - KPUBLAG + 1:
                su = seu
            else:
                v = max(want, sbu)
                while (v - sbu) % PUB:
                    v += 1
                su = v + KPUBLAG - 1
            m = max(m, deb_iter(b - 1, su))
        return m

    worst, at = -10 ** 9, None
    for mv in (0, 8, 24, 40):             # vector reach right / down in samples
        rx = ry = (mv + 15) // 16
        base_g = CHAIN_LAG + KEY_SLACK + slope_g * ry + rx + K_BAND_LAG * ((ry + 1) // BR) + ((mb_h + INTRA_EXTRA) if p_intra else 0)
        for r in range(mb_h):
            for c in range((mb_w + 7) // 8):
                kg = base_g + slope_g * r + 8 * c + K_BAND_LAG * (r // BR)
                m = -10 ** 9
                for x in range(8 * c, min(8 * c + 8, mb_w)):
                    xmax = min(16 * x + 12 + mv + 6, 16 * mb_w - 1)
                    ymax = min(16 * r + 12 + mv + 6, 16 * mb_h - 1)
                    ymin = max(16 * r + mv - 2, 0)
                    xs, yhi = min((xmax + 4) >> 4, mb_w - 1), min((ymax + 4) >> 4, mb_h - 1)
                    ylo = min(max((ymin + 4) >> 4, 0), yhi)
                    bhi, blo = yhi // BR, ylo // BR
                    m = max(m, deb_iter(bhi, xs + L * yhi + 1 + store_lag + 1))
                    if blo != bhi:
                        m = max(m, deb_iter(blo, xs + L * (blo * 16 + 15) + 1 + store_lag + 1))
                if m - kg > worst:
                    worst, at = m - kg, (mv, r, c)
    if verbose:
        print(f"{mb_w}x{mb_h} P {'intra-role' if p_intra else 'deblock-only'} -> successor {'intra-role' if g_intra else 'deblock-only'}: "
              f"largest needed key - key = {worst} at (reach, row, segment) {at}")
    return worst


if __name__ == "__main__":
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (120, 68)
    bad = 0
    for p_intra, g_intra in ((False, False), (False, True), (True, False), (True, True)):
        bad += check(w, h, p_intra, g_intra, verbose=True) >= 0
    sys.exit(1 if bad else 0)
