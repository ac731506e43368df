"""Structural self-checks of the entropy tables (no external H.264 decoder exists in this image, so the
tables are pinned by construction: prefix-freeness, Kraft sums, permutation properties, and equality of the
three independently typed copies: oracle / product / generator)."""
import os
from fractions import Fraction

from util import ROOT, c_array

ORC = os.path.join(ROOT, "oracle", "orc_tables.h")
PROD = os.path.join(ROOT, "jmcodec_amd", "csrc", "h264_cavlc.cpp")
GEN = os.path.join(ROOT, "tools", "h264gen.c")


def codes(lens, bits):
    return [(l, b) for l, b in zip(lens, bits) if l]


def check_prefix_free(cs):
    strs = [format(b, "0%db" % l) for l, b in cs]
    assert all(b < (1 << l) for l, b in cs)
    assert len(set(strs)) == len(strs)
    for i, a in enumerate(strs):
        for j, c in enumerate(strs):
            if i != j:
                assert not c.startswith(a), (a, c)
    return sum(Fraction(1, 1 << l) for l, _ in cs)


def test_coeff_token_tables():
    lens = c_array(ORC, "orc_coeff_token_len")
    bits = c_array(ORC, "orc_coeff_token_bits")
    assert len(lens) == len(bits) == 4 * 68
    for t in range(4):
        cs = codes(lens[68 * t:68 * t + 68], bits[68 * t:68 * t + 68])
        # every legal (total_coeff, trailing_ones) pair has a code: 1 + sum_{tc=1..16} min(tc,3)+1 = 62
        assert len(cs) == 62
        kraft = check_prefix_free(cs)
        assert kraft <= 1
        if t == 3:
            assert all(l == 6 for l, _ in cs)
    # combinations with trailing_ones > total_coeff have no code
    for t in range(4):
        for tc in range(17):
            for t1 in range(4):
                has = lens[68 * t + 4 * tc + t1] != 0
                assert has == (t1 <= tc and t1 <= 3), (t, tc, t1)


def test_chroma_dc_token_table():
    cs = codes(c_array(ORC, "orc_chroma_dc_token_len"), c_array(ORC, "orc_chroma_dc_token_bits"))
    assert len(cs) == 14
    assert check_prefix_free(cs) <= 1


def test_total_zeros_and_run_before_are_complete_codes():
    tzl, tzb = c_array(ORC, "orc_total_zeros_len"), c_array(ORC, "orc_total_zeros_bits")
    for t in range(15):
        cs = codes(tzl[16 * t:16 * t + 16], tzb[16 * t:16 * t + 16])
        assert len(cs) == 16 - t
        k = check_prefix_free(cs)
        # complete prefix codes, except that tzVlcIndex 1 leaves the all-zero 9-bit word unused (start-code safety)
        assert k == (1 - Fraction(1, 512) if t == 0 else 1), (t, k)
    cl, cb = c_array(ORC, "orc_cdc_total_zeros_len"), c_array(ORC, "orc_cdc_total_zeros_bits")
    for t in range(3):
        cs = codes(cl[4 * t:4 * t + 4], cb[4 * t:4 * t + 4])
        assert len(cs) == 4 - t and check_prefix_free(cs) == 1
    rl, rb = c_array(ORC, "orc_run_len"), c_array(ORC, "orc_run_bits")
    for t in range(7):
        cs = codes(rl[15 * t:15 * t + 15], rb[15 * t:15 * t + 15])
        assert len(cs) == (t + 2 if t < 6 else 15)
        k = check_prefix_free(cs)
        assert k == 1 if t < 6 else k == 1 - Fraction(1, 1 << 11)   # run_before > 6: 0000 0000 000 is unused


def test_cbp_mapping_is_a_permutation():
    for name in ("orc_cbp_intra", "orc_cbp_inter"):
        assert sorted(c_array(ORC, name)) == list(range(48))
    assert c_array(ORC, "orc_cbp_intra")[0] == 47 and c_array(ORC, "orc_cbp_inter")[0] == 0


def test_scan_and_deblock_tables():
    assert sorted(c_array(ORC, "orc_zigzag4")) == list(range(16))
    assert sorted(c_array(ORC, "orc_zigzag8")) == list(range(64))
    a, b, tc = c_array(ORC, "orc_alpha"), c_array(ORC, "orc_beta"), c_array(ORC, "orc_tc0")
    assert len(a) == len(b) == 52 and len(tc) == 156
    assert a == sorted(a) and b == sorted(b) and a[15] == 0 and a[16] == 4 and a[51] == 255 and b[51] == 18
    for i in range(52):
        assert tc[3 * i] <= tc[3 * i + 1] <= tc[3 * i + 2]
        if i:
            assert all(tc[3 * i + k] >= tc[3 * i - 3 + k] for k in range(3))
    qpc = c_array(ORC, "orc_qpc_tab")
    assert len(qpc) == 22 and qpc == sorted(qpc) and qpc[0] == 29 and qpc[-1] == 39


def test_three_copies_of_the_tables_agree():
    # product (h264_cavlc.cpp) and generator (h264gen.c) were typed separately from the oracle's header
    o_len, o_bits = c_array(ORC, "orc_coeff_token_len"), c_array(ORC, "orc_coeff_token_bits")
    assert c_array(PROD, "kTokLen") == o_len[:3 * 68] and c_array(PROD, "kTokBits") == o_bits[:3 * 68]
    assert c_array(GEN, "ct_len") == o_len and c_array(GEN, "ct_bits") == o_bits
    assert c_array(PROD, "kCdcLen") == c_array(ORC, "orc_chroma_dc_token_len") == c_array(GEN, "cdc_len")
    assert c_array(PROD, "kCdcBits") == c_array(ORC, "orc_chroma_dc_token_bits") == c_array(GEN, "cdc_bits")
    assert c_array(PROD, "kTzLen") == c_array(ORC, "orc_total_zeros_len") == c_array(GEN, "tz_len")
    assert c_array(PROD, "kTzBits") == c_array(ORC, "orc_total_zeros_bits") == c_array(GEN, "tz_bits")
    assert c_array(PROD, "kCbpIntra") == c_array(ORC, "orc_cbp_intra") == c_array(GEN, "cbp_intra_tab")
    assert c_array(PROD, "kCbpInter") == c_array(ORC, "orc_cbp_inter") == c_array(GEN, "cbp_inter_tab")
    assert c_array(PROD, "kZigzag4") == c_array(ORC, "orc_zigzag4") == c_array(GEN, "zz4")
    kern = os.path.join(ROOT, "jmcodec_amd", "csrc", "kernel_common.h")
    assert c_array(kern, "kAlpha") == c_array(ORC, "orc_alpha") == c_array(GEN, "alpha_tab")
    assert c_array(kern, "kBeta") == c_array(ORC, "orc_beta") == c_array(GEN, "beta_tab")
    assert c_array(kern, "kTc0") == c_array(ORC, "orc_tc0") == c_array(GEN, "tc0_tab")


def test_8x8_scan_and_scaling_tables_agree():
    """zig-zag 8x8 scan: a permutation, each step moves to a neighbouring anti-diagonal position; three typed copies agree."""
    z = c_array(ORC, "orc_zigzag8")
    assert sorted(z) == list(range(64))
    assert z == c_array(PROD, "kZigzag8") == c_array(GEN, "zz8")
    for a, b in zip(z, z[1:]):
        da, db = (a >> 3) + (a & 7), (b >> 3) + (b & 7)
        assert db - da in (0, 1)                      # stays on the anti-diagonal or steps to the next one
    n8 = c_array(ORC, "orc_norm8")
    assert n8 == c_array(GEN, "norm8") and len(n8) == 36
    # normAdjust8x8 grows by 2^(1/6) per qP step: row m+... doubles every 6 (checked across the two ends)
    for c in range(6):
        assert 1.7 < n8[30 + c] / n8[c] < 1.9


def test_cabac_tables_structure():
    """CABAC data tables (generated once from tools/make_cabac_tables.py into the oracle and product headers)."""
    orc = os.path.join(ROOT, "oracle", "orc_cabac_tables.h")
    prod = os.path.join(ROOT, "jmcodec_amd", "csrc", "cabac_tables.h")
    for name_o, name_p in (("orc_cabac_init_mn", "cabac_init_mn"), ("orc_cabac_range_lps", "cabac_range_lps"),
                           ("orc_cabac_trans_lps", "cabac_trans_lps"), ("orc_cabac_sig8_inc", "cabac_sig8_inc"),
                           ("orc_cabac_last8_inc", "cabac_last8_inc")):
        assert c_array(orc, name_o) == c_array(prod, name_p)
    lps = c_array(orc, "orc_cabac_range_lps")
    assert len(lps) == 256
    for s in range(63):                                # LPS range shrinks with the state index and grows with the range quarter
        row, nxt = lps[4 * s:4 * s + 4], lps[4 * s + 4:4 * s + 8]
        assert row == sorted(row) and all(a >= b for a, b in zip(row, nxt))
    # rangeTabLPS follows alpha^s with alpha = (0.01875 / 0.5)^(1/63) within rounding
    alpha = (0.01875 / 0.5) ** (1 / 63)
    for s in range(63):
        for q in range(4):
            ideal = (256 + 64 * q + 32) * 0.5 * alpha ** s          # probability of state s times the middle of range quarter q
            if q == 0:
                ideal = min(ideal, 128.0)
            assert abs(lps[4 * s + q] - ideal) <= max(2.0, 0.03 * ideal), (s, q, lps[4 * s + q], ideal)
    tl = c_array(orc, "orc_cabac_trans_lps")
    assert len(tl) == 64 and tl[0] == 0 and tl[63] == 63 and all(tl[i] <= tl[i + 1] for i in range(62)) and all(tl[i] < i for i in range(1, 63))
    sig8, last8 = c_array(orc, "orc_cabac_sig8_inc"), c_array(orc, "orc_cabac_last8_inc")
    assert len(sig8) == len(last8) == 63 and max(sig8) == 14 and max(last8) == 8 and last8 == sorted(last8)
    mn = c_array(orc, "orc_cabac_init_mn")
    assert len(mn) == 4 * 436 * 2
    for t in range(4):                                 # contexts shared by all slice types start identically
        for ctx in list(range(0, 11)) + list(range(60, 70)):
            assert mn[(t * 436 + ctx) * 2:(t * 436 + ctx) * 2 + 2] == mn[ctx * 2:ctx * 2 + 2]
