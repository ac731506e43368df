"""HEVC host parser of the product (jmcodec_amd/csrc/hevc_*.cpp) against the CPU oracle, without a GPU: the product runs in
parse-only mode and reports an FNV-1a digest over every syntax element it decoded (coding units, prediction units with their final
motion vectors, intra modes, QPs, coefficient levels, SAO parameters); the oracle computes the same digest from its own,
separately written parser (oracle/orc_hevc_ctu.c)."""
import json
import os

import numpy as np
import pytest

import jmcodec_amd
from tools import streams
from test_hevc_oracle import HEVC_CASES
from util import GOLDEN

OPTS = {"parse_only": 1, "digest": 1}


@pytest.fixture(scope="module")
def oracle():
    return streams.OracleHevc()


def product_digest(data, chunks=None):
    with jmcodec_amd.JmAmdDec(1, 1, options=OPTS) as d:
        n = d.decode_stream(data, keep=False, chunks=chunks)
        return d.stat("syntax_digest") & 0xFFFFFFFFFFFFFFFF, d.stat("digest_mbs"), n, d.stat("errors"), jmcodec_amd.jm_nvdec_stream_info(d.h)


@pytest.mark.parametrize("name", sorted(HEVC_CASES))
def test_syntax_digest_equals_oracle(oracle, name):
    kw = HEVC_CASES[name]
    data = streams.generate_hevc(**kw)
    want, ncu = oracle.syntax_digest(data)
    got, cus, n, errors, _ = product_digest(data)
    assert errors == 0
    assert (n, cus) == (kw["frames"], ncu)
    assert got == want, f"{name}: host parser and oracle disagree on the syntax"


def test_golden_digests():
    meta = json.load(open(os.path.join(GOLDEN, "golden_hevc.json")))
    for name, m in meta.items():
        data = open(os.path.join(GOLDEN, name + ".h265"), "rb").read()
        got, cus, n, errors, (w, h) = product_digest(data)
        assert errors == 0 and (n, w, h) == (m["frames"], m["width"], m["height"]) and cus == m["coding_units"]
        assert "%016x" % got == m["syntax_digest"], name


def test_chunking_does_not_matter(oracle):
    """test_player hands whole access units, test_nv_dec one NAL per call; arbitrary byte chunks must work too."""
    data = streams.generate_hevc(**HEVC_CASES["b_gop2"])
    want, _ = oracle.syntax_digest(data)
    rng = np.random.default_rng(3)
    cuts = sorted(set(int(x) for x in rng.integers(1, len(data) - 1, size=25)))
    chunks = [data[a:b] for a, b in zip([0] + cuts, cuts + [len(data)])]
    assert product_digest(data, chunks)[0] == want
    assert product_digest(data, [data])[0] == want


def test_display_order_is_poc_order():
    data = streams.generate_hevc(**HEVC_CASES["b_gop8"])
    with jmcodec_amd.JmAmdDec(1, 1, options=OPTS) as d:
        n = d.decode_stream(data, keep=False)
        pocs = [d.stat(f"display_poc:{i}") for i in range(n)]
    assert pocs == sorted(pocs) and len(set(pocs)) == n == 17


@pytest.mark.parametrize("name", sorted(HEVC_CASES))
def test_display_order_equals_the_oracles(oracle, name):
    """C.5.2 bumping: the product hands the pictures out in the order, and with the PicOrderCntVal, the oracle does -- over every test stream
    (open GOPs with skipped RASL pictures, long-term references, several coded video sequences)."""
    data = streams.generate_hevc(**HEVC_CASES[name])
    want = oracle.display_pocs(data)
    with jmcodec_amd.JmAmdDec(1, 1, options=OPTS) as d:
        n = d.decode_stream(data, keep=False)
        assert [d.stat(f"display_poc:{i}") for i in range(n)] == want


@pytest.mark.parametrize("name", sorted(HEVC_CASES))
def test_job_lists_are_the_ones_the_gpu_tests_confirmed(name):
    """The job lists (coding tree block records, strengths, motion jobs, transform blocks, scaled coefficients) of every test stream, byte for byte
    as they were when the GPU parity tests last confirmed the kernels' output for them (tools/make_hevc_job_digests.py): a change of the host parser
    that is made and timed without a GPU cannot alter what the device receives.  With the syntax digest on, the coefficients of a block are listed
    by position instead of scan order -- the same set."""
    want = json.load(open(os.path.join(GOLDEN, "hevc_job_digests.json")))[name]
    data = streams.generate_hevc(**HEVC_CASES[name])
    with jmcodec_amd.JmAmdDec(1, 1, options={"parse_only": 1, "job_digest": 1}) as d:
        n = d.decode_stream(data, keep=False)
        assert n > 0 and "%016x" % (d.stat("job_digest") & (2 ** 64 - 1)) == want


def test_corrupt_streams_do_not_crash():
    data = streams.generate_hevc(**HEVC_CASES["b_gop2"])
    rng = np.random.default_rng(11)
    for trial in range(40):
        b = bytearray(data)
        for p in rng.integers(40, len(b), size=1 + trial % 5):
            b[p] ^= 1 << int(rng.integers(0, 8))
        if trial % 4 == 0:
            b = b[:int(rng.integers(100, len(b)))]
        with jmcodec_amd.JmAmdDec(1, 1, options={"parse_only": 1}) as d:
            d.decode_stream(bytes(b), keep=False)


def test_h264_and_hevc_handles_side_by_side():
    h264 = streams.generate(width=64, height=48, frames=4, gop=4)
    hevc = streams.generate_hevc(**HEVC_CASES["p_real"])
    with jmcodec_amd.JmAmdDec(0, 1, options={"parse_only": 1}) as a, jmcodec_amd.JmAmdDec(1, 1, options={"parse_only": 1}) as b:
        assert a.decode_stream(h264, keep=False) == 4
        assert b.decode_stream(hevc, keep=False) == HEVC_CASES["p_real"]["frames"]
        assert "H.265" in jmcodec_amd.jm_nvdec_show_dec_info(b.h) and "H.264" in jmcodec_amd.jm_nvdec_show_dec_info(a.h)


def test_hvcc_record_and_length_prefixed_packets(oracle):
    """SURVEY 8f f2 for HEVC: parameter sets through extra_data as an hvcC record, then one length-prefixed packet per access unit"""
    from jmcodec_amd import api
    for name in ("b_gop2", "b_gop8"):          # b_gop8: access units of 256..511 bytes, whose 4-byte length reads 00 00 01 xx like a start code
        data = streams.generate_hevc(**HEVC_CASES[name])
        want, ncu = oracle.syntax_digest(data)
        rec, packets = api.annexb_to_hvcc(data)
        with jmcodec_amd.JmAmdDec(1, 1, options=OPTS, extra_data=rec) as d:
            n = d.decode_stream(None, keep=False, chunks=packets)
            assert (d.stat("syntax_digest") & 0xFFFFFFFFFFFFFFFF, d.stat("digest_mbs"), n) == (want, ncu, HEVC_CASES[name]["frames"])


def test_start_at_cra_and_end_of_sequence(oracle):
    """the product drops the RASL pictures of a CRA that starts decoding (or follows an end-of-sequence NAL) exactly like the oracle"""
    from test_hevc_oracle import cut_at_second_irap
    data = streams.generate_hevc(**HEVC_CASES["open_gop"])
    head, tail = cut_at_second_irap(data)
    for s in (tail, head + b"\x00\x00\x01\x48\x01" + tail):
        want, ncu = oracle.syntax_digest(s)
        got, cus, n, errors, _ = product_digest(s)
        assert errors == 0 and (got, cus) == (want, ncu) and n == oracle.decode(s, 1)[1]


def test_resolution_change_and_repeated_parameter_sets(oracle):
    """a second coded video sequence with another picture size (new SPS at an IDR picture) and parameter sets re-sent in front of every
    IRAP picture: frame counts, sizes and syntax must follow the oracle through the switch"""
    a = streams.generate_hevc(**HEVC_CASES["p_real"])
    b = streams.generate_hevc(**dict(HEVC_CASES["crop_ctb32"], seed=77))
    c = streams.generate_hevc(**HEVC_CASES["open_gop"])
    for data in (a + b, b + a + b, c + a):
        want, ncu = oracle.syntax_digest(data)
        got, cus, n, errors, _ = product_digest(data)
        assert errors == 0 and (got, cus) == (want, ncu) and n == oracle.decode(data, 1)[1]
