"""What pins the entropy-coding tables beyond "three generated copies agree" (VERDICT r1, next-round item 8a).

tools/make_cabac_tables.py and tools/make_hevc_tables.py emit the product's, the oracle's and the generator's copy of every context
table from ONE typed source, so a mistyped initialisation value would be shared by all three and no decode-vs-oracle test could see it.
Two independent checks live here:

1. A SEPARATELY TYPED copy of the widely published part of those tables -- typed for this file per syntax element, in the arrangement the
   public reference decoders print them (per slice type B, P, I for HEVC; per ctxIdx range for H.264), not in the layout of the generated
   headers, and not produced by the scripts.  Equality with all three generated copies is asserted element by element.
2. SHA-256 digests of every table of the product headers with the clause / table number of the standard each one restates
   (tests/golden/table_digests.json, written by tools/make_table_digests.py).  A later round that gets hold of a conformance stream can
   bisect a disagreement to one table; until then the digests make any change to a table a visible, reviewed event.
"""
import hashlib
import json
import os
import re

from util import ROOT, GOLDEN, c_array, c_rows

PROD_CABAC = os.path.join(ROOT, "jmcodec_amd", "csrc", "cabac_tables.h")
ORC_CABAC = os.path.join(ROOT, "oracle", "orc_cabac_tables.h")
PROD_HEVC = os.path.join(ROOT, "jmcodec_amd", "csrc", "hevc_tables.h")
ORC_HEVC = os.path.join(ROOT, "oracle", "orc_hevc_tables.h")
GEN_HEVC = os.path.join(ROOT, "tools", "hevcgen_tables.h")

# ---------------------------------------------------------------------------------------------------------------------------------------
# H.265 9.3.2.2, Tables 9-5 .. 9-37: initValue per syntax element, typed as [B, P, I] rows (the order the HM reference software lists them)
# ---------------------------------------------------------------------------------------------------------------------------------------
CNU = 154
HEVC_INIT = {   # name: (offset macro in hevc_tables.h, [B row, P row, I row])
    "sao_merge_flag":        ("HEVC_CTX_SAO_MERGE", [[153], [153], [153]]),
    "sao_type_idx":          ("HEVC_CTX_SAO_TYPE", [[160], [185], [200]]),
    "split_cu_flag":         ("HEVC_CTX_SPLIT_CU", [[107, 139, 126], [107, 139, 126], [139, 141, 157]]),
    "cu_transquant_bypass":  ("HEVC_CTX_CU_TQ_BYPASS", [[154], [154], [154]]),
    "cu_skip_flag":          ("HEVC_CTX_CU_SKIP", [[197, 185, 201], [197, 185, 201], [CNU, CNU, CNU]]),
    "cu_qp_delta_abs":       ("HEVC_CTX_CU_QP_DELTA", [[154, 154], [154, 154], [154, 154]]),
    "pred_mode_flag":        ("HEVC_CTX_PRED_MODE", [[134], [149], [CNU]]),
    "part_mode":             ("HEVC_CTX_PART_MODE", [[154, 139, 154, 154], [154, 139, 154, 154], [184, CNU, CNU, CNU]]),
    "prev_intra_luma_pred":  ("HEVC_CTX_PREV_INTRA", [[183], [154], [184]]),
    "intra_chroma_pred_mode": ("HEVC_CTX_INTRA_CHROMA", [[152], [152], [63]]),
    "merge_flag":            ("HEVC_CTX_MERGE_FLAG", [[154], [110], [CNU]]),
    "merge_idx":             ("HEVC_CTX_MERGE_IDX", [[137], [122], [CNU]]),
    "inter_pred_idc":        ("HEVC_CTX_INTER_PRED_IDC", [[95, 79, 63, 31, 31], [95, 79, 63, 31, 31], [CNU] * 5]),
    "ref_idx":               ("HEVC_CTX_REF_IDX", [[153, 153], [153, 153], [CNU, CNU]]),
    "abs_mvd_greater0":      ("HEVC_CTX_MVD_G0", [[169], [140], [CNU]]),
    "abs_mvd_greater1":      ("HEVC_CTX_MVD_G1", [[198], [198], [CNU]]),
    "mvp_flag":              ("HEVC_CTX_MVP_FLAG", [[168], [168], [CNU]]),
    "rqt_root_cbf":          ("HEVC_CTX_RQT_ROOT_CBF", [[79], [79], [CNU]]),
    "split_transform_flag":  ("HEVC_CTX_SPLIT_TF", [[224, 167, 122], [124, 138, 94], [153, 138, 138]]),
    "cbf_luma":              ("HEVC_CTX_CBF_LUMA", [[153, 111], [153, 111], [111, 141]]),
    "cbf_cb_cr":             ("HEVC_CTX_CBF_CBCR", [[149, 92, 167, 154], [149, 107, 167, 154], [94, 138, 182, 154]]),
    "transform_skip_flag":   ("HEVC_CTX_TSKIP", [[139, 139], [139, 139], [139, 139]]),
    "last_sig_coeff_prefix": ("HEVC_CTX_LAST_X", [
        [125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93],
        [125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108],
        [110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63]]),
    "last_sig_coeff_y_prefix": ("HEVC_CTX_LAST_Y", None),            # the same initValues as last_sig_coeff_x_prefix (Table 9-27 / 9-28), filled in below
    "coded_sub_block_flag":  ("HEVC_CTX_CSBF", [[121, 140, 61, 154], [121, 140, 61, 154], [91, 171, 134, 141]]),
    "sig_coeff_flag":        ("HEVC_CTX_SIG", [
        [170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
         170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140],
        [155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
         170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140],
        [111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125,
         140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111]]),
    "coeff_abs_level_greater1": ("HEVC_CTX_G1", [
        [154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167, 154, 152, 167, 182],
        [154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182],
        [140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197]]),
    "coeff_abs_level_greater2": ("HEVC_CTX_G2", [[107, 167, 91, 107, 107, 167], [107, 167, 91, 122, 107, 167], [138, 153, 136, 167, 152, 152]]),
}


HEVC_INIT["last_sig_coeff_y_prefix"] = ("HEVC_CTX_LAST_Y", HEVC_INIT["last_sig_coeff_prefix"][1])


def _macro(path, name):
    import re
    m = re.search(r"#define\s+" + name + r"\s+(\d+)", open(path).read())
    assert m, name
    return int(m.group(1))


def test_hevc_context_init_values_against_a_separately_typed_copy():
    covered = set()
    for path, arr in ((PROD_HEVC, "hevc_ctx_init"), (ORC_HEVC, "orch_ctx_init"), (GEN_HEVC, "hg_ctx_init")):
        src = open(path).read()
        import re
        name = next(n for n in (arr, "hevc_ctx_init", "orc_hevc_ctx_init", "hevcgen_ctx_init") if re.search(r"\b" + n + r"\s*\[", src))
        flat = c_array(path, name)
        assert len(flat) == 3 * 154
        table = [flat[i * 154:(i + 1) * 154] for i in range(3)]         # initType 0 (I), 1 (P), 2 (B): 9.3.2.2
        for elem, (macro, rows) in HEVC_INIT.items():
            off = _macro(PROD_HEVC, macro)
            for init_type, row in ((2, rows[0]), (1, rows[1]), (0, rows[2])):
                got = table[init_type][off:off + len(row)]
                assert got == row, f"{os.path.basename(path)}: {elem}, initType {init_type}: {got} != {row}"
                covered.update(range(off, off + len(row)))
    assert covered == set(range(154)), "every context of the table is covered by the separately typed copy"


# ---------------------------------------------------------------------------------------------------------------------------------------
# H.264: every table of the entropy layer against tests/spec_tables_h264.py (typed per syntax element / per code table, see its header)
# ---------------------------------------------------------------------------------------------------------------------------------------
import spec_tables_h264 as spec  # noqa: E402

ORC_TABLES = os.path.join(ROOT, "oracle", "orc_tables.h")
PROD_CAVLC = os.path.join(ROOT, "jmcodec_amd", "csrc", "h264_cavlc.cpp")
PROD_SYNTAX = os.path.join(ROOT, "jmcodec_amd", "csrc", "h264_syntax.cpp")
PROD_KERNEL = os.path.join(ROOT, "jmcodec_amd", "csrc", "kernel_common.h")
GEN_H264 = os.path.join(ROOT, "tools", "h264gen.c")


def _spec_mn_columns():
    """The separately typed (m, n) pairs as four columns [I, idc 0, idc 1, idc 2] x ctxIdx -> pair or None (not defined for that column)."""
    cols = [[None] * 460 for _ in range(4)]
    for table in (spec.CABAC_MN, spec.CABAC_MN_FIELD):
        for elem, (first, by_col) in table.items():
            n = {len(v) for v in by_col.values()}
            assert len(n) == 1, f"{elem}: the columns of one element have one length"
            for key, vals in by_col.items():
                for c in (range(4) if key == "all" else [0 if key == "I" else 1 + key]):
                    for i, mn in enumerate(vals):
                        assert cols[c][first + i] is None, f"{elem}: ctxIdx {first + i} typed twice"
                        cols[c][first + i] = mn
    return cols


def test_the_separately_typed_cabac_copy_covers_every_context_exactly_once():
    cols = _spec_mn_columns()
    for c in range(4):
        have = {i for i in range(460) if cols[c][i] is not None}
        # I slices: no P / B elements (11..59); 276 is end_of_slice_flag (no context variable: decoded with DecodeTerminate)
        want = (set(range(0, 11)) | set(range(60, 276)) | set(range(277, 436))) if c == 0 else (set(range(0, 276)) | set(range(277, 436)))
        assert have == want, (c, sorted(have ^ want))
        for i in have:
            m, n = cols[c][i]
            assert -128 <= m <= 127 and -128 <= n <= 127


def test_h264_cabac_init_values_against_a_separately_typed_copy():
    """ALL frame-coded contexts, ctxIdx 0..435 x {I, cabac_init_idc 0, 1, 2}, product and oracle."""
    cols = _spec_mn_columns()
    for path, name in ((PROD_CABAC, "cabac_init_mn"), (ORC_CABAC, "orc_cabac_init_mn")):
        flat = c_array(path, name)
        n_ctx = len(flat) // 8
        assert len(flat) == 4 * n_ctx * 2 and n_ctx in (436, 460)
        checked = 0
        for t in range(4):
            for i in range(n_ctx):
                want = cols[t][i]
                got = (flat[(t * n_ctx + i) * 2], flat[(t * n_ctx + i) * 2 + 1])
                if want is None:
                    continue
                if 277 <= i <= 398 and got == (0, 0) and n_ctx == 436:
                    continue                                   # field-coded contexts not emitted by this build of the table
                assert got == want, f"{os.path.basename(path)}: ctxIdx {i}, column {('I', 0, 1, 2)[t]}: table has {got}, the separately typed copy {want}"
                checked += 1
        # 0..10, 60..275, 399..435 in every column + 11..59 in the three P / B columns; + 122 x 4 once the table holds the field-coded contexts
        assert checked in (264 + 3 * 313, 264 + 3 * 313 + 4 * 122), checked


def test_h264_cabac_8x8_ctxidxinc_tables():
    for path, pfx in ((PROD_CABAC, "cabac_"), (ORC_CABAC, "orc_cabac_")):
        assert c_array(path, pfx + "sig8_inc") == spec.SIG8_FRAME
        assert c_array(path, pfx + "last8_inc") == spec.LAST8
    assert len(spec.SIG8_FIELD) == 63 and max(spec.SIG8_FIELD) == 14


def test_h264_cavlc_code_tables_against_the_bit_strings_of_the_standard():
    """coeff_token (Table 9-5, all five columns), total_zeros (9-7, 9-8, 9-9a), run_before (9-10): oracle, product, generator."""
    tok = spec.parse_code_table(spec.COEFF_TOKEN, 5)
    assert len(tok) == 62
    for path, ln, bn, ncol in ((ORC_TABLES, "orc_coeff_token_len", "orc_coeff_token_bits", 4), (PROD_CAVLC, "kTokLen", "kTokBits", 3), (GEN_H264, "ct_len",
        "ct_bits", 4)):
        lens, bits = c_array(path, ln), c_array(path, bn)
        assert len(lens) == len(bits) == ncol * 68
        for col in range(ncol):
            for tc in range(17):
                for t1 in range(4):
                    want = tok.get((t1, tc), [None] * 5)[col]
                    got = (lens[col * 68 + 4 * tc + t1], bits[col * 68 + 4 * tc + t1])
                    assert got == (want or (0, 0)), (os.path.basename(path), col, t1, tc, got, want)
    for path, ln, bn in ((ORC_TABLES, "orc_chroma_dc_token_len", "orc_chroma_dc_token_bits"), (PROD_CAVLC, "kCdcLen", "kCdcBits"), (GEN_H264, "cdc_len",
        "cdc_bits")):
        lens, bits = c_array(path, ln), c_array(path, bn)
        for tc in range(5):
            for t1 in range(4):
                want = tok.get((t1, tc), [None] * 5)[4]
                assert (lens[4 * tc + t1], bits[4 * tc + t1]) == (want or (0, 0)), (os.path.basename(path), t1, tc)
    tz = spec.parse_code_table(spec.TOTAL_ZEROS_4x4, 15)
    for path, ln, bn in ((ORC_TABLES, "orc_total_zeros_len", "orc_total_zeros_bits"), (PROD_CAVLC, "kTzLen", "kTzBits"), (GEN_H264, "tz_len", "tz_bits")):
        lens, bits = c_rows(path, ln, 16), c_rows(path, bn, 16)
        assert len(lens) == len(bits) == 15
        for idx in range(15):
            for z in range(16):
                want = tz[(z,)][idx]
                assert (lens[idx][z], bits[idx][z]) == (want or (0, 0)), (os.path.basename(path), idx + 1, z)
    ctz = spec.parse_code_table(spec.TOTAL_ZEROS_CHROMA_DC, 3)
    for path, ln, bn in ((ORC_TABLES, "orc_cdc_total_zeros_len", "orc_cdc_total_zeros_bits"), (PROD_CAVLC, "kCtzLen", "kCtzBits"), (GEN_H264, "ctz_len",
        "ctz_bits")):
        lens, bits = c_rows(path, ln, 4), c_rows(path, bn, 4)
        for idx in range(3):
            for z in range(4):
                want = ctz[(z,)][idx]
                assert (lens[idx][z], bits[idx][z]) == (want or (0, 0)), (os.path.basename(path), idx + 1, z)
    rb = spec.parse_code_table(spec.RUN_BEFORE, 7)
    for path, ln, bn, rows, width in ((ORC_TABLES, "orc_run_len", "orc_run_bits", 7, 15), (GEN_H264, "rb_len", "rb_bits", 7, 15), (PROD_CAVLC, "kRunLen",
        "kRunBits", 6, 7)):
        lens, bits = c_rows(path, ln, width), c_rows(path, bn, width)
        assert len(lens) == len(bits) == rows
        for zl in range(rows):
            for r in range(width):
                want = rb.get((r,), [None] * 7)[zl]
                assert (lens[zl][r], bits[zl][r]) == (want or (0, 0)), (os.path.basename(path), zl + 1, r)
    # zerosLeft > 6 in the product is computed, not tabled: run_before = 7 - (first three bits) or 4 + leading zeros (h264_cavlc.cpp);
    # the column's shape says so
    for r in range(15):
        ln, v = rb[(r,)][6]
        assert (ln, v) == ((3, 7 - r) if r < 7 else (r - 3, 1))


def test_field_scans_in_four_notations():
    """Generator: raster indices; oracle: F4(x, y) / F8(x, y) macros; product: digit pairs "xy"; here: the columns of the figures.  All four must describe
    the same permutation."""
    def from_columns(cols):
        n = len(cols)
        out = [None] * (n * n)
        for x, col in enumerate(cols):
            for y, idx in enumerate(col):
                out[idx] = y * n + x
        assert sorted(out) == list(range(n * n))
        return out
    want4, want8 = from_columns(spec.FIELD_SCAN_4x4_COLUMNS), from_columns(spec.FIELD_SCAN_8x8_COLUMNS)
    assert c_array(GEN_H264, "fs4") == want4 and c_array(GEN_H264, "fs8") == want8
    src = open(ORC_TABLES).read()
    for name, n, want in (("orc_fieldscan4", 4, want4), ("orc_fieldscan8", 8, want8)):
        body = src[src.index(name):]
        body = body[:body.index("};")]
        got = [int(y) * n + int(x) for x, y in re.findall(r"F%d\((\d),\s*(\d)\)" % n, body)]
        assert got == want, name
    src = open(PROD_CAVLC).read()
    for name, n, want in (("xy4", 4, want4), ("xy8", 8, want8)):
        m = re.search(r"\*%s = ((?:\s*\"[0-9 ]+\")+);" % name, src)
        pairs = "".join(re.findall(r'"([0-9 ]+)"', m.group(1))).split()
        assert [int(p[1]) * n + int(p[0]) for p in pairs] == want, name


def test_h264_mapping_scan_and_filter_tables_against_a_separately_typed_copy():
    intra, inter = [a for a, _ in spec.CBP_OF_CODENUM], [b for _, b in spec.CBP_OF_CODENUM]
    assert c_array(ORC_TABLES, "orc_cbp_intra") == c_array(PROD_CAVLC, "kCbpIntra") == c_array(GEN_H264, "cbp_intra_tab") == intra
    assert c_array(ORC_TABLES, "orc_cbp_inter") == c_array(PROD_CAVLC, "kCbpInter") == c_array(GEN_H264, "cbp_inter_tab") == inter
    assert c_array(ORC_TABLES, "orc_zigzag4") == c_array(PROD_CAVLC, "kZigzag4") == c_array(GEN_H264, "zz4") == spec.ZIGZAG_4x4
    assert c_array(ORC_TABLES, "orc_zigzag8") == c_array(PROD_CAVLC, "kZigzag8") == c_array(GEN_H264, "zz8") == spec.ZIGZAG_8x8
    assert c_array(ORC_TABLES, "orc_alpha") == c_array(PROD_KERNEL, "kAlpha") == c_array(GEN_H264, "alpha_tab") == spec.ALPHA
    assert c_array(ORC_TABLES, "orc_beta") == c_array(PROD_KERNEL, "kBeta") == c_array(GEN_H264, "beta_tab") == spec.BETA
    tc0 = [v for row in spec.TC0 for v in row]
    assert len(spec.TC0) == 52 and c_array(ORC_TABLES, "orc_tc0") == c_array(PROD_KERNEL, "kTc0") == c_array(GEN_H264, "tc0_tab") == tc0
    assert c_array(ORC_TABLES, "orc_qpc_tab") == c_array(GEN_H264, "qpc_tab") == spec.QPC_30_51
    assert c_array(ORC_TABLES, "orc_norm4") == c_array(GEN_H264, "norm4") == [v for r in spec.NORM_ADJUST_4x4 for v in r]
    assert c_array(ORC_TABLES, "orc_norm8") == c_array(GEN_H264, "norm8") == [v for r in spec.NORM_ADJUST_8x8 for v in r]
    assert c_array(PROD_SYNTAX, "kDef4Intra") == c_array(GEN_H264, "dflt4_intra") == spec.DEFAULT_4x4_INTRA
    assert c_array(PROD_SYNTAX, "kDef4Inter") == c_array(GEN_H264, "dflt4_inter") == spec.DEFAULT_4x4_INTER
    assert c_array(PROD_SYNTAX, "kDef8Intra") == c_array(GEN_H264, "dflt8_intra") == spec.DEFAULT_8x8_INTRA
    assert c_array(PROD_SYNTAX, "kDef8Inter") == c_array(GEN_H264, "dflt8_inter") == spec.DEFAULT_8x8_INTER


def test_arithmetic_decoder_tables_against_a_separately_typed_copy():
    """rangeTabLPS (first and last rows) and transIdxLPS: H.264 Tables 9-44 / 9-45, H.265 Tables 9-46 / 9-47 (the same engine)."""
    range_first = [[128, 176, 208, 240], [128, 167, 197, 227], [128, 158, 187, 216], [123, 150, 178, 205], [116, 142, 169, 195], [111, 135, 160, 185], [105,
        128, 152, 175], [100, 122, 144, 166]]
    range_last = [[6, 8, 9, 11], [6, 7, 9, 10], [6, 7, 8, 9], [2, 2, 2, 2]]
    trans_lps = [0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63]
    for path, rname, tname in ((PROD_CABAC, "cabac_range_lps", "cabac_trans_lps"), (ORC_CABAC, "orc_cabac_range_lps", "orc_cabac_trans_lps"),
                               (PROD_HEVC, "hevc_range_lps", "hevc_trans_lps")):
        r = c_array(path, rname)
        rows = [r[i * 4:i * 4 + 4] for i in range(64)]
        assert rows[:8] == range_first and rows[60:] == range_last, os.path.basename(path)
        assert [tuple(r) for r in rows] == spec.RANGE_TAB_LPS, os.path.basename(path)          # all 64 rows
        assert c_array(path, tname) == trans_lps, os.path.basename(path)


def test_hevc_constant_tables_against_a_separately_typed_copy():
    """H.265 transform basis, DST, intra angles, interpolation filters, deblocking and chroma-QP tables: typed here from the standard's tables.
    The 32x32 transMatrix is BUILT from its 31 distinct magnitudes and the cosine structure it samples (entry (k, n) = the rounded
    64 * sqrt(2) * cos((2n + 1) k pi / 64), of which the standard lists the integers), not copied as 1024 numbers."""
    mag = {0: 64, 16: 64, 8: 83, 24: 36, 4: 89, 12: 75, 20: 50, 28: 18, 2: 90, 6: 87, 10: 80, 14: 70, 18: 57, 22: 43, 26: 25, 30: 9,
           1: 90, 3: 90, 5: 88, 7: 85, 9: 82, 11: 78, 13: 73, 15: 67, 17: 61, 19: 54, 21: 46, 23: 38, 25: 31, 27: 22, 29: 13, 31: 4}
    want = []
    for k in range(32):
        for n in range(32):
            m = ((2 * n + 1) * k) % 128
            if m > 64:
                m = 128 - m
            want.append(mag[m] if m <= 32 else -mag[64 - m])
    intra_angle = [0, 0] + [32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32]
    inv_angle = [0] * 11 + [-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096] + [0] * 9
    luma = [0, 0, 0, 64, 0, 0, 0, 0, -1, 4, -10, 58, 17, -5, 1, 0, -1, 4, -11, 40, 40, -11, 4, -1, 0, 1, -5, 17, 58, -10, 4, -1]
    chroma = [0, 64, 0, 0, -2, 58, 10, -2, -4, 54, 16, -2, -6, 46, 28, -4, -4, 36, 36, -4, -4, 28, 46, -6, -2, 16, 54, -4, -2, 10, 58, -2]
    beta = [0] * 16 + list(range(6, 19)) + list(range(20, 65, 2))
    tc = [0] * 18 + [1] * 9 + [2] * 4 + [3] * 4 + [4] * 3 + [5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24]
    qpc = list(range(30)) + [29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37] + [q - 6 for q in range(44, 58)]
    dst = [29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29]
    for path, pfx in ((PROD_HEVC, "hevc_"), (ORC_HEVC, None), (GEN_HEVC, None)):
        src = open(path).read()
        import re

        def arr(stem):
            name = next(n for n in (("hevc_" + stem), ("orch_" + stem), ("orc_hevc_" + stem), ("hg_" + stem),
                ("hevcgen_" + stem)) if re.search(r"\b" + n + r"\s*\[", src))
            return c_array(path, name)
        assert arr("trans") == want, os.path.basename(path)
        assert arr("dst") == dst and arr("intra_angle") == intra_angle and arr("inv_angle") == inv_angle, os.path.basename(path)
        assert arr("luma_filter") == luma and arr("chroma_filter") == chroma, os.path.basename(path)
        assert arr("beta_tab") == beta and arr("tc_tab") == tc and arr("qpc_tab") == qpc, os.path.basename(path)
        assert arr("level_scale") == [40, 45, 51, 57, 64, 72], os.path.basename(path)


# ---------------------------------------------------------------------------------------------------------------------------------------
def table_digests():
    """{table: sha256 of its comma-joined integers} for every table of the product headers, with the clause each restates."""
    spec = {
        "cabac_init_mn": (PROD_CABAC, "H.264 9.3.1.1 Tables 9-12..9-23 (frame-coded ctxIdx 0..435; [0] I, [1+cabac_init_idc] P/B)"),
        "cabac_range_lps": (PROD_CABAC, "H.264 9.3.3.2.1.1 Table 9-44 rangeTabLPS"),
        "cabac_trans_lps": (PROD_CABAC, "H.264 9.3.3.2.1.1 Table 9-45 transIdxLPS"),
        "cabac_sig8_inc": (PROD_CABAC, "H.264 9.3.3.1.3 Table 9-43 ctxIdxInc of significant_coeff_flag, 8x8 blocks (frame)"),
        "cabac_last8_inc": (PROD_CABAC, "H.264 9.3.3.1.3 Table 9-43 ctxIdxInc of last_significant_coeff_flag, 8x8 blocks"),
        "hevc_ctx_init": (PROD_HEVC, "H.265 9.3.2.2 Tables 9-5..9-37 initValue, [initType 0 I, 1 P, 2 B][context]"),
        "hevc_range_lps": (PROD_HEVC, "H.265 9.3.4.3.1 Table 9-46 rangeTabLps"),
        "hevc_trans_lps": (PROD_HEVC, "H.265 9.3.4.3.2.2 Table 9-47 transIdxLps"),
        "hevc_trans": (PROD_HEVC, "H.265 8.6.4.2 transMatrix (32x32 DCT basis, equation 8-xxx coefficients)"),
        "hevc_dst": (PROD_HEVC, "H.265 8.6.4.2 4x4 DST-VII matrix"),
        "hevc_intra_angle": (PROD_HEVC, "H.265 8.4.4.2.6 Table 8-4 intraPredAngle"),
        "hevc_inv_angle": (PROD_HEVC, "H.265 8.4.4.2.6 Table 8-5 invAngle"),
        "hevc_luma_filter": (PROD_HEVC, "H.265 8.5.3.3.3.1 Table 8-11 luma interpolation filter fL"),
        "hevc_chroma_filter": (PROD_HEVC, "H.265 8.5.3.3.3.2 Table 8-12 chroma interpolation filter fC"),
        "hevc_beta_tab": (PROD_HEVC, "H.265 8.7.2.5.3 Table 8-12 beta'"),
        "hevc_tc_tab": (PROD_HEVC, "H.265 8.7.2.5.3 Table 8-12 tC'"),
        "hevc_qpc_tab": (PROD_HEVC, "H.265 8.6.1 Table 8-10 QpC as a function of qPi (ChromaArrayType 1)"),
        "hevc_level_scale": (PROD_HEVC, "H.265 8.6.4.2 levelScale"),
        "hevc_scaling_default": (PROD_HEVC, "H.265 7.4.5 Table 7-6 default 8x8 scaling lists (intra, inter)"),
    }
    out = {}
    for name, (path, clause) in spec.items():
        vals = c_array(path, name)
        out[name] = {"restates": clause, "file": os.path.relpath(path, ROOT), "entries": len(vals),
                     "sha256": hashlib.sha256(",".join(map(str, vals)).encode()).hexdigest()}
    return out


def test_table_digests_are_the_committed_ones():
    want = json.load(open(os.path.join(GOLDEN, "table_digests.json")))
    got = table_digests()
    assert got == want, "a constant table changed: review it against the clause it restates, then re-run tools/make_table_digests.py"
