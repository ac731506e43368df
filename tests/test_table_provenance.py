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

from util import ROOT, GOLDEN, c_array

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
# H.264 9.3.1.1, Tables 9-12 .. 9-17: (m, n), typed per ctxIdx range; index 0 = I slices (where defined), 1..3 = cabac_init_idc 0..2
# ---------------------------------------------------------------------------------------------------------------------------------------
H264_MN_ALL = {   # the same for every slice type / cabac_init_idc
    0: [(20, -15), (2, 54), (3, 74), (20, -15), (2, 54), (3, 74), (-28, 127), (-23, 104), (-6, 53), (-1, 54), (7, 51)],            # Table 9-12: mb_type (SI prefix / I), ctxIdx 0..10
    60: [(0, 41), (0, 63), (0, 63), (0, 63), (-9, 83), (4, 86), (0, 97), (-7, 72), (13, 41), (3, 62)],                             # Table 9-17: mb_qp_delta, intra_chroma_pred_mode, prev_intra / rem_intra, ctxIdx 60..69
}
H264_MN_IDC = {   # Table 9-13: mb_skip_flag (P), mb_type (P), sub_mb_type (P), ctxIdx 11..23, per cabac_init_idc
    11: [[(23, 33), (23, 2), (21, 0), (1, 9), (0, 49), (-37, 118), (5, 57), (-13, 78), (-11, 65), (1, 62), (12, 49), (-4, 73), (17, 50)],
         [(22, 25), (34, 0), (16, 0), (-2, 9), (4, 41), (-29, 118), (2, 65), (-6, 71), (-13, 79), (5, 52), (9, 50), (-3, 70), (10, 54)],
         [(29, 16), (25, 0), (14, 0), (-10, 51), (-3, 62), (-27, 99), (26, 16), (-4, 85), (-24, 102), (5, 57), (6, 57), (-17, 73), (14, 57)]],
}
H264_MN_I = {     # Table 9-18 (I slices): mb_field_decoding_flag 70..72, coded_block_pattern luma 73..76
    70: [(0, 11), (1, 55), (0, 69), (-17, 127), (-13, 102), (0, 82), (-7, 74)],
}


def test_h264_cabac_init_values_against_a_separately_typed_copy():
    for path, name in ((PROD_CABAC, "cabac_init_mn"), (ORC_CABAC, "orc_cabac_init_mn")):
        flat = c_array(path, name)
        assert len(flat) == 4 * 436 * 2
        mn = [[(flat[(t * 436 + i) * 2], flat[(t * 436 + i) * 2 + 1]) for i in range(436)] for t in range(4)]
        for first, vals in H264_MN_ALL.items():
            for t in range(4):
                assert mn[t][first:first + len(vals)] == vals, (os.path.basename(path), first, t)
        for first, per_idc in H264_MN_IDC.items():
            for idc in range(3):
                assert mn[1 + idc][first:first + len(per_idc[idc])] == per_idc[idc], (os.path.basename(path), first, idc)
        for first, vals in H264_MN_I.items():
            assert mn[0][first:first + len(vals)] == vals, (os.path.basename(path), first)


def test_arithmetic_decoder_tables_against_a_separately_typed_copy():
    """rangeTabLPS (first and last rows) and transIdxLPS: H.264 Tables 9-44 / 9-45, H.265 Tables 9-46 / 9-47 (the same engine)."""
    range_first = [[128, 176, 208, 240], [128, 167, 197, 227], [128, 158, 187, 216], [123, 150, 178, 205], [116, 142, 169, 195], [111, 135, 160, 185], [105, 128, 152, 175], [100, 122, 144, 166]]
    range_last = [[6, 8, 9, 11], [6, 7, 9, 10], [6, 7, 8, 9], [2, 2, 2, 2]]
    trans_lps = [0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63]
    for path, rname, tname in ((PROD_CABAC, "cabac_range_lps", "cabac_trans_lps"), (ORC_CABAC, "orc_cabac_range_lps", "orc_cabac_trans_lps"),
                               (PROD_HEVC, "hevc_range_lps", "hevc_trans_lps")):
        r = c_array(path, rname)
        rows = [r[i * 4:i * 4 + 4] for i in range(64)]
        assert rows[:8] == range_first and rows[60:] == range_last, os.path.basename(path)
        assert c_array(path, tname) == trans_lps, os.path.basename(path)


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
