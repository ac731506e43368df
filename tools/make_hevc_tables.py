#!/usr/bin/env python3
"""Emit the constant tables of ITU-T H.265 (Main profile) as C headers:

    oracle/orc_hevc_tables.h             (CPU oracle, prefix orch_)
    tools/hevcgen_tables.h               (stream generator, prefix hg_)
    jmcodec_amd/csrc/hevc_tables.h       (product: host entropy decoder + HIP kernels, prefix hevc_)

Contents: CABAC context initialisation values (Tables 9-5 .. 9-37, the three initType columns), the context layout,
the 32x32 core transform matrix (8.6.4.2, generated from its 31 distinct magnitudes by the cosine index structure), the 4x4
DST matrix, intra prediction angles (Table 8-4 / 8-5), interpolation filters (Tables 8-11 / 8-12), beta / tC (Table 8-12
of 8.7.2.5.3), the chroma QP mapping (Table 8-10) and the default scaling lists (Tables 7-5 / 7-6).
rangeTabLps / transIdxLps are the H.264 tables (tools/make_cabac_tables.py).

PARITY NOTE: no third-party HEVC stream or decoder exists in this image, so these numbers are restated from the published
standard without an external check ("parity unpinned", oracle/orc_hevc.h).  tests/test_hevc_tables.py checks what structure
offers: orthogonality of the transform basis, filter taps summing to 64, monotone beta / tC, table lengths.
"""
import os

CNU = 154
# name: (count, initType0 (I), initType1, initType2)
CTX = [
    ("SAO_MERGE", [153], [153], [153]),
    ("SAO_TYPE", [200], [185], [160]),
    ("SPLIT_CU", [139, 141, 157], [107, 139, 126], [107, 139, 126]),
    ("CU_TQ_BYPASS", [154], [154], [154]),
    ("CU_SKIP", [CNU] * 3, [197, 185, 201], [197, 185, 201]),
    ("CU_QP_DELTA", [154, 154], [154, 154], [154, 154]),
    ("PRED_MODE", [CNU], [149], [134]),
    ("PART_MODE", [184, CNU, CNU, CNU], [154, 139, 154, 154], [154, 139, 154, 154]),
    ("PREV_INTRA", [184], [154], [183]),
    ("INTRA_CHROMA", [63], [152], [152]),
    ("MERGE_FLAG", [CNU], [110], [154]),
    ("MERGE_IDX", [CNU], [122], [137]),
    ("INTER_PRED_IDC", [CNU] * 5, [95, 79, 63, 31, 31], [95, 79, 63, 31, 31]),
    ("REF_IDX", [CNU] * 2, [153, 153], [153, 153]),
    ("MVD_G0", [CNU], [140], [169]),
    ("MVD_G1", [CNU], [198], [198]),
    ("MVP_FLAG", [CNU], [168], [168]),
    ("RQT_ROOT_CBF", [CNU], [79], [79]),
    ("SPLIT_TF", [153, 138, 138], [124, 138, 94], [224, 167, 122]),
    ("CBF_LUMA", [111, 141], [153, 111], [153, 111]),
    ("CBF_CBCR", [94, 138, 182, 154], [149, 107, 167, 154], [149, 92, 167, 154]),
    ("TSKIP", [139, 139], [139, 139], [139, 139]),
    ("LAST_X", [110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63],
               [125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108],
               [125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93]),
    ("LAST_Y", None, None, None),          # same values as LAST_X
    ("CSBF", [91, 171, 134, 141], [121, 140, 61, 154], [121, 140, 61, 154]),
    ("SIG", [111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125,
             140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111],
            [155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
             170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140],
            [170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
             170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140]),
    ("G1", [140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197],
           [154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182],
           [154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167, 154, 152, 167, 182]),
    ("G2", [138, 153, 136, 167, 152, 152], [107, 167, 91, 122, 107, 167], [107, 167, 91, 107, 107, 167]),
]

ODD32 = [90, 90, 88, 85, 82, 78, 73, 67, 61, 54, 46, 38, 31, 22, 13, 4]      # cos(m*pi/64), m odd
ODD16 = [90, 87, 80, 70, 57, 43, 25, 9]                                       # m = 2 mod 4
ODD8 = [89, 75, 50, 18]                                                       # m = 4 mod 8


def cmag(m):
    """integer magnitude standing for 64*sqrt(2)*cos(m*pi/64), 0 <= m <= 32"""
    if m == 0 or m == 16:
        return 64
    if m == 32:
        return 0
    if m & 1:
        return ODD32[(m - 1) // 2]
    if m % 4 == 2:
        return ODD16[(m - 2) // 4]
    if m % 8 == 4:
        return ODD8[(m - 4) // 8]
    return {8: 83, 24: 36}[m]


def coef(k, n):
    m = (k * (2 * n + 1)) % 128
    if m > 64:
        m = 128 - m
    return cmag(m) if m <= 32 else -cmag(64 - m)


TRANS = [[coef(k, n) for n in range(32)] for k in range(32)]
DST = [[29, 55, 74, 84], [74, 74, 0, -74], [84, -29, -74, 55], [55, -84, 74, -29]]
ANGLE = [0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32]
INV_ANGLE = {11: -4096, 12: -1638, 13: -910, 14: -630, 15: -482, 16: -390, 17: -315, 18: -256, 19: -315, 20: -390, 21: -482, 22: -630, 23: -910, 24: -1638,
    25: -4096}
LUMA_F = [[0, 0, 0, 64, 0, 0, 0, 0], [-1, 4, -10, 58, 17, -5, 1, 0], [-1, 4, -11, 40, 40, -11, 4, -1], [0, 1, -5, 17, 58, -10, 4, -1]]
CHROMA_F = [[0, 64, 0, 0], [-2, 58, 10, -2], [-4, 54, 16, -2], [-6, 46, 28, -4], [-4, 36, 36, -4], [-4, 28, 46, -6], [-2, 16, 54, -4], [-2, 10, 58, -2]]
BETA = [0] * 16 + [6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64]
TC = [0] * 18 + [1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24]
QPC = list(range(30)) + [29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37] + [q - 6 for q in range(44, 58)]
SL_INTRA = [16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 17, 16, 17, 16, 17, 18, 17, 18, 18, 17, 18, 21, 19, 20, 21, 20, 19, 21, 24, 22, 22, 24,
            24, 22, 22, 24, 25, 25, 27, 30, 27, 25, 25, 29, 31, 35, 35, 31, 29, 36, 41, 44, 41, 36, 47, 54, 54, 47, 65, 70, 65, 88, 88, 115]
SL_INTER = [16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 17, 17, 17, 17, 17, 18, 18, 18, 18, 18, 18, 20, 20, 20, 20, 20, 20, 20, 24, 24, 24, 24,
            24, 24, 24, 24, 25, 25, 25, 25, 25, 25, 25, 28, 28, 28, 28, 28, 28, 33, 33, 33, 33, 33, 41, 41, 41, 41, 54, 54, 54, 71, 71, 91]
LEVEL_SCALE = [40, 45, 51, 57, 64, 72]


def layout():
    off, names, cols = 0, [], [[], [], []]
    last = None
    for name, a, b, c in CTX:
        if a is None:
            a, b, c = last
        assert len(a) == len(b) == len(c), name
        names.append((name, off, len(a)))
        off += len(a)
        for col, v in zip(cols, (a, b, c)):
            col.extend(v)
        last = (a, b, c)
    return names, cols, off


def arr(vals, per=24):
    vals = [str(v) for v in vals]
    return ",\n    ".join(", ".join(vals[i:i + per]) for i in range(0, len(vals), per))


def emit(path, prefix, guard):
    import importlib.util
    spec = importlib.util.spec_from_file_location("h264tab", os.path.join(os.path.dirname(os.path.abspath(__file__)), "make_cabac_tables.py"))
    h264 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(h264)
    range_lps = [v for pair in h264.RANGE_LPS for v in pair]
    names, cols, n = layout()
    P = prefix.upper()
    o = [f"// generated by tools/make_hevc_tables.py -- do not edit.  ITU-T H.265 constant tables (see the script for provenance).",
         f"#ifndef {guard}", f"#define {guard}", "#include <stdint.h>", ""]
    for name, off, cnt in names:
        o.append(f"#define {P}CTX_{name} {off}   /* {cnt} */")
    o.append(f"#define {P}N_CTX {n}")
    o.append("")
    o.append(f"static const uint8_t {prefix}ctx_init[3][{n}] = {{")
    for c in cols:
        o.append("  { " + arr(c) + " },")
    o.append("};")
    rows = ["{" + ", ".join(map(str, range_lps[4 * i:4 * i + 4])) + "}" for i in range(64)]
    o.append(f"static const uint8_t {prefix}range_lps[64][4] = {{\n    " + ",\n    ".join(", ".join(rows[r:r + 4]) for r in range(0, 64, 4)) + "\n};")
    o.append(f"static const uint8_t {prefix}trans_lps[64] = {{\n    " + arr(h264.TRANS_LPS, 32) + "\n};")
    o.append(f"static const int8_t {prefix}trans[32][32] = {{")
    for r in TRANS:
        o.append("  { " + ", ".join(f"{v:3d}" for v in r) + " },")
    o.append("};")
    o.append(f"static const int8_t {prefix}dst[4][4] = {{ " + ", ".join("{" + ", ".join(map(str, r)) + "}" for r in DST) + " };")
    o.append(f"static const int8_t {prefix}intra_angle[35] = {{ " + ", ".join(map(str, ANGLE)) + " };")
    o.append(f"static const int16_t {prefix}inv_angle[35] = {{ " + ", ".join(str(INV_ANGLE.get(i, 0)) for i in range(35)) + " };")
    o.append(f"static const int8_t {prefix}luma_filter[4][8] = {{ " + ", ".join("{" + ", ".join(map(str, r)) + "}" for r in LUMA_F) + " };")
    o.append(f"static const int8_t {prefix}chroma_filter[8][4] = {{ " + ", ".join("{" + ", ".join(map(str, r)) + "}" for r in CHROMA_F) + " };")
    o.append(f"static const uint8_t {prefix}beta_tab[52] = {{ " + ", ".join(map(str, BETA)) + " };")
    o.append(f"static const uint8_t {prefix}tc_tab[54] = {{ " + ", ".join(map(str, TC)) + " };")
    o.append(f"static const uint8_t {prefix}qpc_tab[58] = {{ " + ", ".join(map(str, QPC)) + " };")
    o.append(f"static const uint8_t {prefix}level_scale[6] = {{ " + ", ".join(map(str, LEVEL_SCALE)) + " };")
    o.append(f"static const uint8_t {prefix}scaling_default[2][64] = {{\n  {{ " + arr(SL_INTRA, 32) + " },\n  { " + arr(SL_INTER, 32) + " } };")
    o.append("#endif")
    with open(path, "w") as f:
        f.write("\n".join(o) + "\n")


if __name__ == "__main__":
    assert len(BETA) == 52 and len(TC) == 54 and len(QPC) == 58 and len(SL_INTRA) == 64 and len(SL_INTER) == 64
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emit(os.path.join(root, "oracle", "orc_hevc_tables.h"), "orch_", "ORC_HEVC_TABLES_H")
    emit(os.path.join(root, "tools", "hevcgen_tables.h"), "hg_", "HEVCGEN_TABLES_H")
    emit(os.path.join(root, "jmcodec_amd", "csrc", "hevc_tables.h"), "hevc_", "JMAMD_HEVC_TABLES_H")
    print("contexts:", layout()[2])
