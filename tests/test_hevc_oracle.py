"""HEVC CPU oracle (oracle/orc_hevc_*.c) against the generator's independently written reconstruction (tools/hevcgen.c), the
committed golden vectors and structural properties of the constant tables.  CPU only.

PARITY NOTE (oracle/orc_hevc.h): the reference ships no HEVC fixture and this image holds no third-party HEVC stream or
decoder, so these tests pin the oracle against the second code path and against regressions -- "parity unpinned" w.r.t. the
standard's tables.
"""
import json
import os
import tempfile

import numpy as np
import pytest

from tools import streams
from util import GOLDEN, ROOT, c_array, md5

HEVC_CASES = {
    "intra": dict(width=64, height=64, frames=1),
    "intra_ctb16_fuzz": dict(width=80, height=48, frames=2, intra_period=1, ctb_log2=4, mode=1, seed=2),
    "p_real": dict(width=96, height=80, frames=4, num_ref=2),
    "p_fuzz_refs": dict(width=96, height=80, frames=6, num_ref=4, mode=1, seed=3, merge_cand=2),
    "b_gop2": dict(width=96, height=80, frames=7, gop=2, num_ref=2, mode=1, seed=4),
    "b_gop8": dict(width=128, height=96, frames=17, gop=8, num_ref=2, seed=5),
    "crop_ctb32": dict(width=90, height=70, frames=3, ctb_log2=5, mode=1, seed=6),
    "min_cb16": dict(width=96, height=80, frames=3, ctb_log2=5, min_cb_log2=4, mode=1, seed=7),
    "no_filters": dict(width=96, height=80, frames=3, sao=0, deblock=0, tmvp=0, amp=0, mode=1, seed=8),
    "deblock_override": dict(width=96, height=80, frames=3, deblock=2, slice_ctus=3, ctb_log2=4, mode=1, seed=9),
    "tskip_sdh": dict(width=96, height=80, frames=3, tskip=1, sdh=1, mode=1, seed=10, qp=22),
    "dqp_depths": dict(width=128, height=96, frames=3, dqp=4, mode=1, seed=11, cb_qp_off=-5, cr_qp_off=7),
    "pcm_bypass": dict(width=96, height=80, frames=3, pcm=1, bypass=1, mode=1, seed=12),
    "pcm_filtered": dict(width=96, height=80, frames=2, pcm=2, mode=1, seed=13),
    "pcm_only_8bit": dict(width=96, height=80, frames=3, pcm=3, ctb_log2=4, min_cb_log2=4, sao=0, deblock=0, gop=0, num_ref=1, seed=21),
    # the known-answer stream below
    "cip": dict(width=96, height=80, frames=4, cip=1, mode=1, seed=14),
    "wp_b": dict(width=96, height=80, frames=7, gop=2, num_ref=2, wp=1, mode=1, seed=15),
    "rplm": dict(width=96, height=80, frames=6, num_ref=3, rplm=1, mode=1, seed=16),
    "long_term": dict(width=64, height=64, frames=40, num_ref=2, lt_ref=1, seed=17),
    "scaling_default": dict(width=96, height=80, frames=3, scaling=1, mode=1, seed=18),
    "scaling_sps": dict(width=96, height=80, frames=3, scaling=2, mode=1, seed=19),
    "scaling_pps": dict(width=96, height=80, frames=3, scaling=3, mode=1, seed=20),
    "wpp": dict(width=128, height=96, frames=3, wpp=1, ctb_log2=4, mode=1, seed=21),
    "tiles": dict(width=128, height=96, frames=3, tile_cols=3, tile_rows=2, ctb_log2=4, mode=1, seed=22),
    "tiles_no_lf": dict(width=128, height=96, frames=3, tile_cols=2, tile_rows=2, ctb_log2=4, mode=1, seed=23),
    "slices_dep": dict(width=128, height=96, frames=3, slice_ctus=5, dep_slices=1, ctb_log2=4, mode=1, seed=24),
    "slices_wpp_dep": dict(width=128, height=96, frames=3, slice_ctus=8, dep_slices=1, wpp=1, ctb_log2=4, mode=1, seed=25),
    "cabac_init": dict(width=96, height=80, frames=5, gop=1, cabac_init=2, mode=1, seed=26),
    "par_merge": dict(width=96, height=80, frames=4, par_mrg=4, mode=1, seed=27),
    "rps_sps_gop8": dict(width=96, height=80, frames=17, gop=8, num_ref=2, rps_sps=1, mode=1, seed=29),
    "rps_sps_p": dict(width=96, height=80, frames=9, num_ref=3, rps_sps=1, mode=1, seed=30),
    "open_gop": dict(width=96, height=80, frames=20, gop=2, num_ref=2, open_gop=1, intra_period=6, rps_sps=1, seed=31),
    # dependent slice segments that OPEN a tile: the first CTB of a tile starts from initialised context variables, not from the previous segment's (9.3.1)
    "tiles_dep_slices": dict(width=128, height=96, frames=3, tile_cols=2, tile_rows=2, slice_ctus=4, dep_slices=1, ctb_log2=4, mode=1, seed=40),
    "tiles_explicit": dict(width=128, height=96, frames=3, tile_cols=3, tile_rows=2, ctb_log2=4, mode=1, seed=34),      # seed & 2: uniform_spacing_flag = 0
    "poc_wrap": dict(width=64, height=64, frames=70, intra_period=70, gop=2, num_ref=2, seed=36),                          # MaxPicOrderCntLsb = 32 < 70
    "small_tb": dict(width=96, height=80, frames=3, max_tb_log2=3, depth_inter=1, depth_intra=0, mode=1, seed=28),
}


@pytest.fixture(scope="module")
def oracle():
    return streams.OracleHevc()


@pytest.mark.parametrize("name", sorted(HEVC_CASES))
def test_oracle_equals_generator_reconstruction(oracle, name):
    """Two separately written code paths (decoder from the syntax, encoder's own reconstruction loop) must agree bit-exactly."""
    with tempfile.NamedTemporaryFile(suffix=".yuv") as tf:
        data = streams.generate_hevc(recon_path=tf.name, **HEVC_CASES[name])
        recon = open(tf.name, "rb").read()
    out, n, w, h = oracle.decode(data, 1)
    assert n == HEVC_CASES[name]["frames"] and (w, h) == (HEVC_CASES[name]["width"], HEVC_CASES[name]["height"])
    assert out == recon, f"{name}: oracle output differs from the generator's reconstruction"


def test_cases_cover_the_tools(oracle):
    seen = {}
    for name in ("b_gop2", "tskip_sdh", "pcm_bypass", "wp_b", "rplm", "long_term", "wpp", "tiles", "tiles_dep_slices", "slices_dep", "dqp_depths", "b_gop8",
        "cip"):
        for k, v in oracle.tools(streams.generate_hevc(**HEVC_CASES[name])).items():
            seen[k] = seen.get(k, 0) + v
    for tool in ("intra_cu", "skip_cu", "merge_pu", "amvp_pu", "bi_pu", "amp", "nxn", "tu4", "tu8", "tu16", "tu32", "dst", "sign_hiding", "transform_skip",
        "tq_bypass", "pcm",
                 "cu_qp_delta", "sao_band", "sao_edge", "weighted_pred", "tmvp", "wpp_rows", "tiles", "dependent_slices", "long_term_ref", "rplm", "b_slices",
                 "dependent_segment_opens_tile",   # 9.3.1: initialised contexts, not the stored ones (a second shared misreading, found in round 3)
                 # 8.5.3.2.3: B0 / B2 dropped as duplicates of a B1 that is available but was itself pruned against A1 (all three programs once misread it)
                 "merge_b0_b2_vs_pruned_b1"):
        assert seen.get(tool, 0) > 0, f"no test stream exercises {tool}"


def test_golden_vectors(oracle):
    meta = json.load(open(os.path.join(GOLDEN, "golden_hevc.json")))
    assert len(meta) >= 6
    for name, m in meta.items():
        data = open(os.path.join(GOLDEN, name + ".h265"), "rb").read()
        out, n, w, h = oracle.decode(data, 1)
        assert (n, w, h) == (m["frames"], m["width"], m["height"])
        assert md5(out) == m["md5_i420"], name
        nv12, _, _, _ = oracle.decode(data, 0)
        assert md5(nv12) == m["md5_nv12"], name
        dig, ncu = oracle.syntax_digest(data)
        assert "%016x" % dig == m["syntax_digest"] and ncu == m["coding_units"], name
        assert streams.generate_hevc(**m["params"]) == data, f"{name}: the generator no longer reproduces the committed stream"


def test_nv12_and_i420_outputs_hold_the_same_samples(oracle):
    data = streams.generate_hevc(**HEVC_CASES["crop_ctb32"])
    i420, n, w, h = oracle.decode(data, 1)
    nv12, _, _, _ = oracle.decode(data, 0)
    fs = w * h * 3 // 2
    for k in range(n):
        a = np.frombuffer(i420[k * fs:(k + 1) * fs], np.uint8); b = np.frombuffer(nv12[k * fs:(k + 1) * fs], np.uint8)
        assert (a[:w * h] == b[:w * h]).all()
        uv = b[w * h:].reshape(h // 2, w // 2, 2)
        assert (uv[:, :, 0].ravel() == a[w * h:w * h + w * h // 4]).all() and (uv[:, :, 1].ravel() == a[w * h + w * h // 4:]).all()


def test_corrupt_streams_fail_cleanly(oracle):
    data = bytearray(streams.generate_hevc(**HEVC_CASES["b_gop2"]))
    rng = np.random.default_rng(5)
    for _ in range(40):
        b = bytearray(data)
        for p in rng.integers(60, len(b), size=3):
            b[p] ^= 1 << int(rng.integers(0, 8))
        try:
            oracle.decode(bytes(b), 1)
        except RuntimeError:
            pass


# ---------------------------------------------------------------------------------------------------------
# structure of the constant tables (tools/make_hevc_tables.py): the only checks available without a third-party stream
# ---------------------------------------------------------------------------------------------------------
TABLES = os.path.join(ROOT, "oracle", "orc_hevc_tables.h")


def test_transform_basis_structure():
    t = np.array(c_array(TABLES, "orch_trans")).reshape(32, 32)
    assert (t[0] == 64).all()
    for k in range(32):                                   # even / odd symmetry of the DCT-II basis functions
        assert (t[k] == (-1) ** k * t[k][::-1]).all()
    g = t @ t.T
    assert np.abs(g - np.diag(g.diagonal())).max() < 400 and abs(g.diagonal() - 64 * 64 * 32).max() < 400     # near-orthogonal, norm 2^17
    for n in (4, 8, 16):                                  # the smaller transforms are the even rows of the larger
        sub = t[::32 // n, :n]
        gs = sub @ sub.T
        assert np.abs(gs - np.diag(gs.diagonal())).max() < 400
    d = np.array(c_array(TABLES, "orch_dst")).reshape(4, 4)
    gd = d @ d.T
    assert np.abs(gd - np.diag(gd.diagonal())).max() <= 16 and abs(gd.diagonal() - 128 * 128).max() < 100        # DST-VII basis, norm 2^14


def test_filters_and_loop_filter_tables():
    lf = np.array(c_array(TABLES, "orch_luma_filter")).reshape(4, 8)
    cf = np.array(c_array(TABLES, "orch_chroma_filter")).reshape(8, 4)
    assert (lf.sum(1) == 64).all() and (cf.sum(1) == 64).all()
    assert (lf[1] == lf[3][::-1]).all() and (lf[2] == lf[2][::-1]).all()
    for k in range(1, 8):
        assert (cf[k] == cf[8 - k][::-1]).all()
    beta, tc = c_array(TABLES, "orch_beta_tab"), c_array(TABLES, "orch_tc_tab")
    assert len(beta) == 52 and len(tc) == 54 and beta == sorted(beta) and tc == sorted(tc)
    assert beta[15] == 0 and beta[16] == 6 and beta[51] == 64 and tc[17] == 0 and tc[18] == 1 and tc[53] == 24
    qpc = c_array(TABLES, "orch_qpc_tab")
    assert qpc[:30] == list(range(30)) and qpc[43] == 37 and qpc[44] == 38 and qpc[57] == 51 and qpc == sorted(qpc)
    ang = c_array(TABLES, "orch_intra_angle")
    assert ang[2] == 32 and ang[10] == 0 and ang[18] == -32 and ang[26] == 0 and ang[34] == 32 and all(ang[18 + k] == ang[18 - k] for k in range(17))
    inv = c_array(TABLES, "orch_inv_angle")
    for m in range(11, 26):
        assert abs(inv[m] - round(256 * 32 / ang[m])) <= 1


def test_cabac_tables_shape_and_identity_between_copies():
    init = np.array(c_array(TABLES, "orch_ctx_init")).reshape(3, 154)
    assert init.min() >= 0 and init.max() <= 255
    # the three copies (oracle, generator, product) come from one script and must stay identical
    for path, prefix in ((os.path.join(ROOT, "tools", "hevcgen_tables.h"), "hg_"), (os.path.join(ROOT, "jmcodec_amd", "csrc", "hevc_tables.h"), "hevc_")):
        assert c_array(path, prefix + "ctx_init") == init.ravel().tolist()
        assert c_array(path, prefix + "trans") == c_array(TABLES, "orch_trans")
    h264 = os.path.join(ROOT, "oracle", "orc_cabac_tables.h")
    assert c_array(TABLES, "orch_range_lps") == c_array(h264, "orc_cabac_range_lps") and c_array(TABLES, "orch_trans_lps") == c_array(h264,
        "orc_cabac_trans_lps")


def nal_types(data):
    import jmcodec_amd
    return [(n.lstrip(b"\x00")[1] >> 1) & 63 for n in jmcodec_amd.split_nalus(data)]


def cut_at_second_irap(data):
    """the stream from its second IRAP picture on (with the parameter sets the generator repeats there)"""
    import jmcodec_amd
    nalus = jmcodec_amd.split_nalus(data)
    types = [(n.lstrip(b"\x00")[1] >> 1) & 63 for n in nalus]
    iraps = [i for i, t in enumerate(types) if 16 <= t <= 21]
    k = iraps[1]
    while types[k - 1] in (32, 33, 34):
        k -= 1
    return b"".join(nalus[:k]), b"".join(nalus[k:])


def test_open_gop_stream_has_cra_and_rasl_pictures(oracle):
    data = streams.generate_hevc(**HEVC_CASES["open_gop"])
    t = nal_types(data)
    assert t.count(21) >= 2 and t.count(8) >= 4 and t.count(19) == 1


def test_decoding_from_a_cra_drops_its_rasl_pictures(oracle):
    """8.1.3: NoRaslOutputFlag = 1 for a CRA picture that starts the bitstream; its RASL pictures are not output."""
    kw = HEVC_CASES["open_gop"]
    data = streams.generate_hevc(**kw)
    full, n_full, w, h = oracle.decode(data, 1)
    head, tail = cut_at_second_irap(data)
    out, n, _, _ = oracle.decode(tail, 1)
    fs = w * h * 3 // 2
    # pictures of the tail in display order: the CRA is frame 6; frames 4, 5 (its RASL pictures) are gone, everything from 6 on stays identical
    assert n == kw["frames"] - 6
    assert out == full[6 * fs:]
    # end of sequence NAL in front of the CRA: same effect inside one stream (POC MSB restarts, RASL pictures dropped)
    eos = b"\x00\x00\x01\x48\x01"
    out2, n2, _, _ = oracle.decode(head + eos + tail, 1)
    n_head = oracle.decode(head, 1)[1]
    assert n2 == n_head + n and out2[n_head * fs:] == out


def test_pcm_known_answer(oracle):
    """A true known answer, independent of the reconstruction code of either decoder: in a stream whose coding units are all pcm_flag = 1 with 8-bit
    samples, no in-loop filter and 16x16 coding tree blocks, every decoded block must literally be a run of payload bytes of the slice data, in
    decoding order (7.3.8.7 pcm_sample: luma 16x16, then Cb 8x8, then Cr 8x8)."""
    import jmcodec_amd
    from util import unescape
    w, h, frames = 96, 80, 3
    data = streams.generate_hevc(**HEVC_CASES["pcm_only_8bit"])
    out, n, ow, oh = oracle.decode(data, 1)
    assert (n, ow, oh) == (frames, w, h)
    slices = []
    for x in jmcodec_amd.split_nalus(data):
        nal = x.lstrip(b"\x00")[1:]                    # after the start code
        if ((nal[0] >> 1) & 63) < 32:                  # VCL NAL units
            slices.append(unescape(nal[2:]))
    assert len(slices) == frames
    fs = w * h * 3 // 2
    for f, rbsp in enumerate(slices):
        fr = np.frombuffer(out[f * fs:(f + 1) * fs], np.uint8)
        Y = fr[:w * h].reshape(h, w); U = fr[w * h:w * h * 5 // 4].reshape(h // 2, w // 2); V = fr[w * h * 5 // 4:].reshape(h // 2, w // 2)
        pos = 0
        for cy in range(h // 16):
            for cx in range(w // 16):
                payload = (Y[cy * 16:cy * 16 + 16, cx * 16:cx * 16 + 16].tobytes() + U[cy * 8:cy * 8 + 8, cx * 8:cx * 8 + 8].tobytes()
                           + V[cy * 8:cy * 8 + 8, cx * 8:cx * 8 + 8].tobytes())
                k = rbsp.find(payload, pos)
                assert k >= 0, (f, cx, cy)
                # between two payloads: the arithmetic decoder's 9 + 7 initialisation bits, end_of_slice_segment_flag, (cu_skip_flag, pred_mode_flag,)
                # part_mode,
                # pcm_flag and the alignment bits -- a few bytes; before the first one also the slice segment header
                assert k - pos <= (24 if (cx, cy) == (0, 0) else 6), (f, cx, cy, k - pos)
                pos = k + 384
        assert len(rbsp) - pos <= 4                    # end_of_slice_segment_flag + rbsp_slice_segment_trailing_bits
