import hashlib
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
SESSION_NOTES = []      # lines a test wants in the terminal summary whatever the verbosity (conftest.py prints them)


def golden_meta():
    return json.load(open(os.path.join(GOLDEN, "golden.json")))


def golden_stream(name):
    return open(os.path.join(GOLDEN, name + ".h264"), "rb").read()


def md5(b):
    return hashlib.md5(b).hexdigest()


def c_array(path, name):
    """Extract a (possibly nested) integer array initialiser called `name` from a C/C++ source as a flat list."""
    src = open(path).read()
    m = re.search(r"\b" + re.escape(name) + r"\s*(\[[^\]]*\]\s*)+=\s*\{", src)
    assert m, f"{name} not found in {path}"
    i = m.end() - 1
    depth, j = 0, i
    while True:
        if src[j] == "{":
            depth += 1
        elif src[j] == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    body = re.sub(r"/\*.*?\*/", "", src[i:j + 1], flags=re.S)
    return [int(x, 0) for x in re.findall(r"-?\b(?:0x[0-9a-fA-F]+|\d+)\b", body)]


def c_rows(path, name, width):
    """A two-dimensional integer array initialiser as a list of rows, each padded with zeros to `width` (C semantics of short inner braces)."""
    src = open(path).read()
    m = re.search(r"\b" + re.escape(name) + r"\s*(\[[^\]]*\]\s*)+=\s*\{", src)
    assert m, f"{name} not found in {path}"
    i = m.end()
    body = re.sub(r"/\*.*?\*/", "", src[i:src.index(";", i)], flags=re.S)
    rows = [[int(x, 0) for x in re.findall(r"-?\b(?:0x[0-9a-fA-F]+|\d+)\b", r)] for r in re.findall(r"\{([^{}]*)\}", body)]
    assert rows and all(len(r) <= width for r in rows), (name, width)
    return [r + [0] * (width - len(r)) for r in rows]


def unescape(nal):
    out = bytearray()
    z = 0
    for b in nal:
        if z >= 2 and b == 3:
            z = 0
            continue
        out.append(b)
        z = z + 1 if b == 0 else 0
    return bytes(out)


# the cases every parity test runs (generator keyword arguments)
PARITY_CASES = {
    "pcm": dict(width=64, height=48, frames=2, pcm_only=1, gop=2, deblock=0),
    "intra_nodb": dict(width=64, height=48, frames=1, gop=1, deblock=0),
    "intra_db": dict(width=64, height=48, frames=2, gop=1, deblock=1),
    "p_nodb": dict(width=64, height=48, frames=4, gop=4, deblock=0),
    "p_db": dict(width=64, height=48, frames=4, gop=4, deblock=1),
    "fuzz_nodb": dict(width=96, height=80, frames=6, gop=6, mode=1, deblock=0, seed=77),
    "fuzz_multiref_slices": dict(width=96, height=80, frames=8, gop=4, mode=1, deblock=1, num_ref=3, slices=2, seed=77),
    "fuzz_poc0_nonref_idc2": dict(width=96, height=80, frames=12, gop=6, mode=1, num_ref=4, slices=3, seed=78, poc_type=0, nonref_period=3, deblock=2),
    "fuzz_cip_offsets": dict(width=176, height=144, frames=8, gop=8, mode=1, num_ref=2, seed=79, cip=1, chroma_qp_off=-3, alpha_off=2, beta_off=-2),
    "fuzz_crop_odd_mbs": dict(width=90, height=70, frames=6, gop=3, mode=1, num_ref=2, seed=80),
    "real_qvga": dict(width=320, height=240, frames=10, gop=10, seed=5),
    "real_lowqp": dict(width=128, height=96, frames=5, gop=5, seed=6, qp=12),
    "real_highqp": dict(width=128, height=96, frames=5, gop=5, seed=7, qp=44),
    "one_mb": dict(width=16, height=16, frames=4, gop=2, mode=1, seed=81),
    "wide_strip": dict(width=640, height=16, frames=3, gop=3, mode=1, seed=82),
    "tall_strip": dict(width=16, height=400, frames=3, gop=3, mode=1, seed=83),
    # Main / High profile: CABAC (all three cabac_init_idc tables), 8x8 transform, Intra8x8
    "cabac_pcm": dict(width=48, height=32, frames=2, pcm_only=1, gop=2, cabac=1, seed=3),
    "cabac_real": dict(width=176, height=144, frames=6, gop=6, seed=21, cabac=1),
    "cabac_fuzz_idc1_multiref": dict(width=96, height=80, frames=8, gop=4, mode=1, num_ref=3, slices=2, seed=84, cabac=1, cabac_idc=1),
    "cabac_fuzz_idc2_cip": dict(width=96, height=80, frames=8, gop=8, mode=1, num_ref=2, slices=3, seed=85, cabac=1, cabac_idc=2, cip=1, deblock=2,
    chroma_qp_off=2),
    "high_cabac_fuzz": dict(width=96, height=80, frames=8, gop=4, mode=1, num_ref=2, seed=86, cabac=1, t8x8=1),
    "high_cavlc_fuzz": dict(width=96, height=80, frames=8, gop=4, mode=1, num_ref=2, seed=87, t8x8=1, poc_type=0),
    "high_real_qvga": dict(width=320, height=240, frames=8, gop=8, seed=22, cabac=1, t8x8=1, qp=30),
    "high_crop_odd_mbs": dict(width=90, height=70, frames=6, gop=3, mode=1, seed=88, cabac=1, cabac_idc=1, t8x8=1, slices=2),
}

# B pictures (Main / High): direct spatial / temporal, bi-prediction, implicit and explicit weights, CAVLC and CABAC
B_CASES = {
    # explicit weighted prediction in P slices
    "wp_explicit_p": dict(width=96, height=80, frames=8, gop=8, mode=1, num_ref=3, seed=89, wp=1),
    "wp_explicit_p_cabac": dict(width=96, height=80, frames=6, gop=6, mode=1, num_ref=2, seed=90, wp=1, cabac=1, t8x8=1),
    "b_real_spatial": dict(width=176, height=144, frames=10, gop=10, seed=91, bframes=2),
    "b_real_temporal_cabac": dict(width=176, height=144, frames=10, gop=10, seed=92, bframes=2, direct_temporal=1, cabac=1),
    "b_fuzz_cavlc": dict(width=96, height=80, frames=10, gop=10, mode=1, seed=93, bframes=2, num_ref=3),
    "b_fuzz_cabac_high": dict(width=96, height=80, frames=12, gop=6, mode=1, seed=94, bframes=3, num_ref=4, cabac=1, t8x8=1, slices=2),
    "b_fuzz_temporal_noinf8": dict(width=96, height=80, frames=10, gop=10, mode=1, seed=95, bframes=2, num_ref=3, direct_temporal=1, dinf8=0, cabac=1,
    cabac_idc=1),
    "b_fuzz_spatial_noinf8": dict(width=96, height=80, frames=10, gop=10, mode=1, seed=96, bframes=1, dinf8=0, deblock=2, slices=3, cip=1),
    "b_fuzz_implicit_wp": dict(width=96, height=80, frames=10, gop=10, mode=1, seed=97, bframes=2, num_ref=3, wp=2, cabac=1, cabac_idc=2),
    "b_fuzz_explicit_wp": dict(width=96, height=80, frames=10, gop=10, mode=1, seed=98, bframes=2, num_ref=2, wp=1, t8x8=1),
    "b_crop_odd_mbs": dict(width=90, height=70, frames=8, gop=8, mode=1, seed=99, bframes=3, num_ref=2, cabac=1),
    # scaling matrices (SPS / PPS lists, fall-back rules, default tables)
    "scaling_sps_cavlc": dict(width=96, height=80, frames=6, gop=6, mode=1, seed=100, scaling=1),
    "scaling_pps_high_cabac": dict(width=96, height=80, frames=8, gop=4, mode=1, seed=101, scaling=2, t8x8=1, cabac=1, num_ref=2),
    "scaling_sps_high_b": dict(width=96, height=80, frames=8, gop=8, mode=1, seed=102, scaling=1, t8x8=1, cabac=1, bframes=2),
    "scaling_real_qvga": dict(width=320, height=240, frames=6, gop=6, seed=103, scaling=1, t8x8=1, cabac=1),
    # reference list modification (8.2.4.3), memory management control operations and long-term references (8.2.5.4)
    "rplm_p": dict(width=96, height=80, frames=14, gop=14, mode=1, seed=104, rplm=1, num_ref=4),
    "rplm_b_cabac": dict(width=96, height=80, frames=14, gop=14, mode=1, seed=105, rplm=1, num_ref=4, bframes=2, cabac=1),
    "rplm_b_temporal": dict(width=96, height=80, frames=12, gop=12, mode=1, seed=106, rplm=1, num_ref=3, bframes=1, direct_temporal=1),
    "mmco_longterm": dict(width=96, height=80, frames=20, gop=20, mode=1, seed=107, mmco=1, num_ref=4),
    "mmco_longterm_rplm_cabac": dict(width=96, height=80, frames=20, gop=10, mode=1, seed=108, mmco=1, rplm=1, num_ref=3, cabac=1, t8x8=1),
    "mmco_two_refs_poc0": dict(width=96, height=80, frames=20, gop=20, mode=1, seed=17, mmco=1, num_ref=2, poc_type=0, slices=2),
    "mmco_one_ref": dict(width=96, height=80, frames=16, gop=16, mode=1, seed=109, mmco=1, num_ref=1),
    # frame_mbs_only_flag = 0 without MBAFF, every picture a frame picture (field_pic_flag = 0): an interlace-capable stream coded progressively.
    # pic_height_in_map_units counts field macroblock rows, the vertical crop unit is four luma rows, the slice header carries field_pic_flag
    "fmo0_cavlc_fuzz": dict(width=96, height=96, frames=8, gop=4, mode=1, num_ref=2, slices=2, seed=110, fmo0=1),
    "fmo0_cabac_b_crop": dict(width=176, height=156, frames=10, gop=10, mode=1, seed=111, fmo0=1, cabac=1, t8x8=1, bframes=2, num_ref=3),
    "fmo0_real": dict(width=320, height=256, frames=6, gop=6, seed=112, fmo0=1, cabac=1),
    # picture order count (8.2.1): type 1 (expected-delta cycle, offset_for_non_ref_pic, delta_pic_order_cnt[0 / 1]), bottom-field deltas
    # (PicOrderCnt = Min(top, bottom)) in P / B streams, and memory management operation 5 (frame_num and order counts restart) with types 0, 1, 2
    "poc1_cycle_nonref": dict(width=96, height=80, frames=16, gop=16, mode=1, seed=113, poc_type=1, nonref_period=3, num_ref=2),
    "poc1_bottom_cabac": dict(width=96, height=80, frames=12, gop=6, mode=1, seed=114, poc_type=1, poc_bottom=1, nonref_period=2, num_ref=3, cabac=1, slices=2),
    "poc0_bottom_b_temporal": dict(width=96, height=80, frames=13, gop=13, mode=1, seed=115, poc_bottom=1, bframes=2, num_ref=3, direct_temporal=1, cabac=1),
    "poc0_bottom_b_implicit": dict(width=96, height=80, frames=13, gop=13, mode=1, seed=116, poc_bottom=1, bframes=3, num_ref=3, wp=2),
    "mmco5_poc0_bottom_nonref": dict(width=64, height=48, frames=70, gop=70, mode=1, seed=2, poc_type=0, poc_bottom=1, nonref_period=2, mmco=2, num_ref=3),
    "mmco5_poc1": dict(width=96, height=80, frames=24, gop=24, mode=1, seed=117, poc_type=1, poc_bottom=1, nonref_period=3, mmco=2, num_ref=3),
    "mmco5_poc2_cabac": dict(width=96, height=80, frames=24, gop=24, mode=1, seed=118, poc_type=2, nonref_period=3, mmco=2, num_ref=2, cabac=1),
}
# Interlace, picture-adaptive (P-only streams): every I / P picture a frame or two field pictures (paff=1), or always two fields (paff=2); the first field of
# either parity.  Field scans, field contexts, alternating-parity lists, field picture numbers in list modification and marking, the sliding window
# across first / second fields, frame pictures that meet half-marked stores, explicit weights by field index, all three order-count types.
PAFF_CASES = {
    "paff_fields_cavlc": dict(width=96, height=96, frames=8, gop=8, seed=201, paff=2, num_ref=2),
    "paff_fields_cabac": dict(width=96, height=96, frames=8, gop=8, seed=202, paff=2, num_ref=2, cabac=1, cabac_idc=1),
    "paff_adaptive_fuzz": dict(width=96, height=64, frames=14, gop=9, mode=1, seed=204, paff=1, num_ref=3, slices=2, rplm=1, mmco=1),
    "paff_adaptive_fuzz_cabac_wp": dict(width=80, height=96, frames=14, gop=7, mode=1, seed=204, paff=1, num_ref=4, cabac=1, rplm=1, mmco=2, wp=1, poc_type=0,
                                        nonref_period=3),
    "paff_poc1_t8x8_scaling": dict(width=96, height=96, frames=12, gop=12, mode=1, seed=205, paff=1, num_ref=3, t8x8=1, scaling=1, poc_type=1, poc_bottom=1,
    deblock=2,
                                   slices=3, chroma_qp_off=-3, alpha_off=2, beta_off=-2),
    "paff_poc2_cip_crop": dict(width=90, height=88, frames=10, gop=5, mode=1, seed=206, paff=2, num_ref=2, cip=1, poc_type=2, cabac=1, cabac_idc=2),
    "paff_real_qvga": dict(width=320, height=224, frames=6, gop=6, seed=207, paff=1, num_ref=2, cabac=1),
    # B field pictures (every picture of the stream two fields: the colocated field of a B field is then always a field picture, 8.4.1.2.1 One_To_One)
    "paff_b_spatial": dict(width=96, height=96, frames=10, gop=10, mode=1, seed=211, paff=2, bframes=2, num_ref=3, slices=2),
    "paff_b_temporal_cabac": dict(width=96, height=64, frames=13, gop=13, mode=1, seed=212, paff=2, bframes=3, num_ref=3, cabac=1, direct_temporal=1, rplm=1),
    "paff_b_implicit_wp_t8x8": dict(width=80, height=96, frames=10, gop=10, mode=1, seed=213, paff=2, bframes=2, num_ref=4, wp=2, t8x8=1, direct_temporal=1),
    "paff_b_explicit_wp_real": dict(width=176, height=160, frames=7, gop=7, seed=214, paff=2, bframes=1, num_ref=2, wp=1, cabac=1, cabac_idc=2),
    # frame and field pictures mixed around B pictures: the colocated picture of a B field may be a frame picture (Frm_To_Fld), that of a B frame a field
    # pair (Fld_To_Frm) -- 8.4.1.2.1 Tables 8-6 / 8-8, vertical vectors halved / doubled in temporal direct prediction
    "paff_mixed_b_spatial": dict(width=96, height=96, frames=13, gop=13, mode=1, seed=215, paff=1, bframes=2, num_ref=3, slices=2),
    "paff_mixed_b_temporal_cabac": dict(width=96, height=64, frames=13, gop=13, mode=1, seed=216, paff=1, bframes=3, num_ref=3, cabac=1, direct_temporal=1,
    rplm=1),
    "paff_mixed_b_implicit_wp": dict(width=80, height=96, frames=14, gop=7, mode=1, seed=217, paff=1, bframes=2, num_ref=4, wp=2, t8x8=1, direct_temporal=1),
    "paff_mixed_b_real": dict(width=176, height=160, frames=10, gop=10, seed=218, paff=1, bframes=2, num_ref=2, cabac=1),
}
# gaps_in_frame_num_value_allowed_flag = 1 and frame_num values that no picture carries (8.2.5.2): the decoder infers the frames that were not sent -- they pass
# through the sliding window (older pictures leave earlier than they otherwise would) and sit in the initial lists, shifting the indices of the real ones
GAPS_CASES = {
    "gaps_fuzz_multiref": dict(width=96, height=80, frames=20, gop=20, mode=1, num_ref=3, seed=221, gaps=1, rplm=1),
    "gaps_mmco_cabac_poc1": dict(width=96, height=80, frames=24, gop=12, mode=1, num_ref=3, seed=222, gaps=1, mmco=1, cabac=1, poc_type=1),
    "gaps_real_nonref_poc2": dict(width=128, height=96, frames=16, gop=16, seed=223, gaps=1, num_ref=2, nonref_period=3, poc_type=2),
    "gaps_paff_cabac": dict(width=96, height=96, frames=16, gop=16, mode=1, seed=224, gaps=1, paff=1, num_ref=3, cabac=1),
    # redundant_pic_cnt_present_flag = 1 with slices of redundant coded pictures behind the primary ones (their payload is not slice data): dropped
    "redundant_slices_baseline": dict(width=96, height=80, frames=8, gop=8, mode=1, num_ref=2, slices=2, seed=225, redundant=1),
    "redundant_slices_cabac_b": dict(width=96, height=80, frames=9, gop=9, mode=1, num_ref=2, seed=226, redundant=1, cabac=1, bframes=2, gaps=0),
}
ALL_CASES = dict(PARITY_CASES)
ALL_CASES.update(B_CASES)
ALL_CASES.update(PAFF_CASES)
ALL_CASES.update(GAPS_CASES)
