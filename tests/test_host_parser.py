"""Host side of the product (no GPU needed): Annex-B splitting, header parsing, DPB / display order and the CAVLC
job builder, compared with the CPU oracle through the macroblock syntax digest (parse_only mode produces no pixels)."""
import os
import random

import pytest

import jmcodec_amd
from jmcodec_amd import api
from tools import streams
from util import ALL_CASES as PARITY_CASES, golden_meta, golden_stream


def _product_digest(data, chunks=None):
    with api.JmAmdDec(0, 1, options={"parse_only": 1, "digest": 1}) as d:
        n = d.decode_stream(data, keep=False, chunks=chunks)
        pocs = [d.stat(f"display_poc:{i}") for i in range(n)]
        return d.stat("syntax_digest") & (2 ** 64 - 1), d.stat("digest_mbs"), n, d.stat("errors"), pocs, d.stat("job_bytes")


@pytest.mark.parametrize("name", sorted(PARITY_CASES))
def test_syntax_digest_matches_oracle(oracle, name):
    data = streams.generate(**PARITY_CASES[name])
    od, on = oracle.syntax_digest(data)
    pd, pn, frames, errors, _, job_bytes = _product_digest(data)
    assert errors == 0
    assert frames == PARITY_CASES[name]["frames"]
    assert (pd, pn) == (od, on)
    assert job_bytes > 0


@pytest.mark.parametrize("name", sorted(golden_meta()))
def test_golden_streams_parse(oracle, name):
    data = golden_stream(name)
    od, on = oracle.syntax_digest(data)
    pd, pn, frames, errors, _, _ = _product_digest(data)
    assert (pd, pn, frames, errors) == (od, on, golden_meta()[name]["frames"], 0)


def test_chunking_does_not_matter(oracle):
    """jm_nvdec_decode_frame accepts any chunk: one NAL per call (test_nv_dec.cpp:186-215), whole access units
    (test_player.cpp:253), the whole file, or arbitrary byte runs must all give the same pictures."""
    data = golden_stream("ip_fuzz_crop_90x70")
    ref = _product_digest(data)
    assert _product_digest(data, chunks=[data])[:4] == ref[:4]
    rng = random.Random(1)
    for _ in range(3):
        cuts, i = [], 0
        while i < len(data):
            n = rng.choice([1, 2, 3, 5, 17, 100, 1000])
            cuts.append(data[i:i + n]); i += n
        assert _product_digest(data, chunks=cuts)[:4] == ref[:4]


def test_display_order_is_poc_order_within_each_idr_period(oracle):
    kw = dict(width=96, height=80, frames=12, gop=6, mode=1, num_ref=4, slices=3, seed=78, poc_type=0, nonref_period=3, deblock=2)
    _, _, n, _, pocs, _ = _product_digest(streams.generate(**kw))
    assert n == 12
    assert pocs[:6] == sorted(pocs[:6]) and pocs[6:] == sorted(pocs[6:])
    assert pocs[0] == 0 and pocs[6] == 0


@pytest.mark.parametrize("kw", [dict(poc_bottom=1, nonref_period=2, mmco=2, num_ref=3), dict(nonref_period=3, mmco=2, num_ref=2),
                                dict(poc_bottom=1, bframes=2), dict(poc_bottom=1, nonref_period=3), dict(mmco=1, num_ref=4)])
@pytest.mark.parametrize("seed", [2, 4, 7, 14])
def test_picture_order_counts_equal_the_encoders_intent(oracle, kw, seed):
    """8.2.1.1 known answer: the generator does not reconstruct order counts, it CHOOSES TopFieldOrderCnt (2 per picture from the last IDR picture or
    operation 5), a bottom delta in -1..1, and writes their low bits; oracle and product must rebuild exactly Min(top, bottom) for every frame --
    through lsb wraps (70 pictures against 6- or 8-bit lsb), non-reference pictures between a picture with operation 5 and the next one (the msb / lsb
    pair of the previous REFERENCE picture applies), and a TopFieldOrderCnt that stays above 0 after operation 5 when the bottom field lies lower."""
    cfg = dict(width=32, height=32, frames=70, gop=70, mode=1, seed=seed, poc_type=0)
    cfg.update(kw)
    data = streams.generate(**cfg)
    want = streams.last_pocs()
    assert len(want) == 70 and want[0] == 0
    assert oracle.display_pocs(data) == want
    with jmcodec_amd.JmAmdDec(0, 1, options={"parse_only": 1}) as d:
        n = len(d.decode_stream(data))
        assert [d.stat(f"display_poc:{i}") for i in range(n)] == want


@pytest.mark.parametrize("poc_type", [1, 2])
def test_order_counts_of_types_1_and_2_rise_in_display_order(oracle, poc_type):
    """Types 1 and 2 derive the counts from frame_num (8.2.1.2 / 8.2.1.3): the values differ from the encoder's own, the ORDER may not -- and both
    decoders must agree on every value; after operation 5 the picture's own count is 0 and the following ones start again above it."""
    data = streams.generate(width=32, height=32, frames=60, gop=60, mode=1, seed=12, poc_type=poc_type, poc_bottom=1, nonref_period=3, mmco=2, num_ref=3)
    intent = streams.last_pocs()
    got = oracle.display_pocs(data)
    with jmcodec_amd.JmAmdDec(0, 1, options={"parse_only": 1}) as d:
        n = len(d.decode_stream(data))
        assert [d.stat(f"display_poc:{i}") for i in range(n)] == got and n == 60
    restarts = [i for i in range(60) if intent[i] == 0]
    assert len(restarts) >= 3, "the stream holds pictures with operation 5"
    for a, b in zip(restarts, restarts[1:] + [60]):
        assert got[a] == 0 and all(got[i] < got[i + 1] for i in range(a, b - 1)), (a, b, got[a:b])


def test_cabac_engine_shortcuts_equal_the_bin_by_bin_decoder(tmp_path):
    """n bypass bins as one reciprocal multiplication, a unary prefix by counting leading zeros (h264_cabac.h): a C++ check against the
    one-bin-at-a-time operations of 9.3.3.2.3 (tests/cabac_engine_check.cpp)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ is missing")
    exe = str(tmp_path / "cabac_engine_check")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cabac_engine_check.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mbmi", "-mbmi2", "-mlzcnt", "-o", exe, src])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_empty_and_garbage_input():
    with api.JmAmdDec(0, 1, options={"parse_only": 1}) as d:
        assert d.decode_stream(b"", keep=False) == 0
    with api.JmAmdDec(0, 1, options={"parse_only": 1}) as d:
        assert d.decode_stream(bytes(range(256)) * 8, keep=False) == 0
    # a truncated stream must not crash and must still display the complete pictures before the cut
    data = golden_stream("ip_real_96x80")
    with api.JmAmdDec(0, 1, options={"parse_only": 1}) as d:
        n = d.decode_stream(data[:len(data) * 2 // 3], keep=False)
        assert 1 <= n <= 6


def test_api_protocol_and_info_block():
    """got_frame / is_exit / output_frame conventions of nv_dec.cpp:406-478 and :750-828, info block of :663-683."""
    import ctypes as C
    import re
    data = golden_stream("ip_real_96x80")
    h = api.jm_nvdec_create_handle()
    api.lib().jm_amddec_set_option(h, b"parse_only", 1)
    api.lib().jm_amddec_set_option(h, b"sync", 1)
    assert api.jm_nvdec_init(0, 1, None, 0, h) == 0
    buf = C.create_string_buffer(96 * 80 * 3 // 2)
    assert api.jm_nvdec_output_frame(buf, len(buf), h)[0] == -1            # no frame yet
    frames = 0
    for nal in api.split_nalus(data):
        ret, got = api.jm_nvdec_decode_frame(nal, len(nal), h)
        assert ret == 0 and got in (0, 1)
        if got:
            frames += 1
            assert api.jm_nvdec_output_frame(buf, 10, h)[0] == -2           # buffer too small
            ret, n = api.jm_nvdec_output_frame(buf, len(buf), h)
            assert ret == n == 96 * 80 * 3 // 2                             # returns the size, not 0 (nv_dec.cpp:827)
    assert api.jm_nvdec_stream_info(h) == (96, 80)
    assert not api.jm_nvdec_is_exit(h)
    while not api.jm_nvdec_is_exit(h):
        ret, got = api.jm_nvdec_decode_frame(None, 0, h)
        frames += got
    assert frames == 6
    info = api.jm_nvdec_show_dec_info(h)
    assert re.fullmatch(r"=+\nCodec:\t\tH\.264\nDisplay:\t96 x 80\nPixel Format:\tYV12\nFrame Count:\t6\nElapsed Time:\t\d+ ms\nDecode FPS:\t[\d.]+ "
        r"fps\n=+\n", info), info
    # once EOS was sent further input is ignored (nv_dec.cpp:374-375)
    ret, got = api.jm_nvdec_decode_frame(data, len(data), h)
    assert (ret, got) == (0, 0)
    api.jm_nvdec_deinit(h)


def test_thirdparty_cabac_stream_parses_like_the_oracle(oracle):
    """High profile / CABAC / 8x8 transform: the host entropy decoder and the oracle agree on every syntax element of the
    third-party clip (tests/golden/README.md)."""
    import os
    data = open(os.path.join(os.path.dirname(__file__), "golden", "thirdparty_realshort.h264"), "rb").read()
    od, on = oracle.syntax_digest(data)
    pd, pn, frames, errors, pocs, _ = _product_digest(data)
    assert (pd, pn, frames, errors) == (od, on, 36, 0)
    assert on == 36 * 20 * 15


def test_avcc_extradata_and_length_prefixed_packets(oracle):
    """parameter sets through jm_nvdec_init(extra_data) as an avcC record, packets as length-prefixed NAL units
    (what FFmpeg hands test_player for MP4 input, test_player.cpp:221-226, nv_dec.cpp:334-360) == the Annex-B decode."""
    import os
    for data in (golden_stream("ip_fuzz_96x80"), open(os.path.join(os.path.dirname(__file__), "golden", "thirdparty_realshort.h264"), "rb").read()):
        want = oracle.syntax_digest(data)
        for ls in (4, 2):
            rec, packets = api.annexb_to_avcc(data, ls)
            with api.JmAmdDec(0, 1, options={"parse_only": 1, "digest": 1}, extra_data=rec) as d:
                n = d.decode_stream(b"", keep=False, chunks=packets)
                assert (d.stat("syntax_digest") & (2 ** 64 - 1), d.stat("digest_mbs")) == want
                assert d.stat("errors") == 0 and n > 0
        # Annex-B parameter sets as extra_data, slices afterwards
        nal = api.split_nalus(data)
        head = b"".join(x for x in nal if (x.lstrip(b"\x00")[1] & 31) in (7, 8))
        rest = [x for x in nal if (x.lstrip(b"\x00")[1] & 31) not in (7, 8)]
        with api.JmAmdDec(0, 1, options={"parse_only": 1, "digest": 1}, extra_data=head) as d:
            d.decode_stream(b"", keep=False, chunks=rest)
            assert (d.stat("syntax_digest") & (2 ** 64 - 1), d.stat("digest_mbs")) == want


def test_resolution_change_between_sequences_host_side():
    """A new SPS with another picture size at an IDR picture starts a new coded video sequence (the reference re-creates its decoder in
    the sequence callback, nv_dec.cpp:23-30): all frames of all sequences come out, in order, none lost."""
    a = streams.generate(width=96, height=80, frames=5, gop=5, mode=1, seed=1, num_ref=2)
    b = streams.generate(width=176, height=144, frames=4, gop=4, mode=1, seed=2, cabac=1)
    c = streams.generate(width=64, height=48, frames=3, gop=3, mode=1, seed=3, bframes=1)
    with api.JmAmdDec(0, 1, options={"parse_only": 1}) as d:
        frames = d.decode_stream(a + b + c)
        assert [len(f) for f in frames] == [96 * 80 * 3 // 2] * 5 + [176 * 144 * 3 // 2] * 4 + [64 * 48 * 3 // 2] * 3
        assert d.stat("errors") == 0


# ---- hostile parameter sets: every ue(v) field is range-checked before it is narrowed ------------------------------
class _Bits:
    def __init__(self):
        self.b = []

    def u(self, n, v):
        self.b += [(v >> (n - 1 - i)) & 1 for i in range(n)]
        return self

    def ue(self, v):
        v += 1
        n = v.bit_length()
        return self.u(n - 1, 0).u(n, v)

    def se(self, v):
        return self.ue(2 * v - 1 if v > 0 else -2 * v)

    def nal(self, header):
        bits = self.b + [1]
        bits += [0] * (-len(bits) % 8)
        raw = bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))
        out, z = bytearray(), 0
        for x in raw:                                   # emulation prevention (7.4.1)
            if z >= 2 and x <= 3:
                out.append(3); z = 0
            out.append(x); z = z + 1 if x == 0 else 0
        return b"\x00\x00\x00\x01" + bytes([header]) + bytes(out)


def _sps(log2_fn=0, poc_type=2, log2_lsb=0, num_ref=1, mb_w1=5, mb_h1=4, crop=None, profile=66, frame_mbs_only=1):
    b = _Bits().u(8, profile).u(8, 0).u(8, 40).ue(0)
    b.ue(log2_fn).ue(poc_type)
    if poc_type == 0:
        b.ue(log2_lsb)
    b.ue(num_ref).u(1, 0).ue(mb_w1).ue(mb_h1).u(1, frame_mbs_only)
    if not frame_mbs_only:
        b.u(1, 0)
    b.u(1, 1)
    if crop:
        b.u(1, 1)
        for c in crop:
            b.ue(c)
    else:
        b.u(1, 0)
    b.u(1, 0)
    return b.nal(0x67)


def _pps(nref0=0, init_qp=0, cqo=0):
    return _Bits().ue(0).ue(0).u(1, 0).u(1, 0).ue(0).ue(nref0).ue(0).u(1, 0).u(2, 0).se(init_qp).se(0).se(cqo).u(1, 1).u(1, 0).u(1, 0).nal(0x68)


def _idr_slice():
    # first_mb 0, slice_type 7 (I), pps 0, frame_num u(4) 0, idr_pic_id 0, [poc type 2], no_output_of_prior 0, long_term 0, qp_delta 0, deblock idc 1
    return _Bits().ue(0).ue(7).ue(0).u(4, 0).ue(0).u(1, 0).u(1, 0).se(0).ue(1).u(16, 0xFFFF).nal(0x65)


BIG = 0xFFFFFFFE
HOSTILE_SPS = {
    "mb_w_wraps_negative": dict(mb_w1=BIG, mb_h1=2),
    "mb_h_wraps_negative": dict(mb_w1=2, mb_h1=BIG),
    "both_wrap": dict(mb_w1=0xFFFFFBFF, mb_h1=BIG),
    "mb_count_too_large": dict(mb_w1=1023, mb_h1=1023),
    "log2_max_frame_num_60": dict(log2_fn=60),
    "log2_max_poc_lsb_50": dict(poc_type=0, log2_lsb=50),
    "poc_type_3": dict(poc_type=3),
    "num_ref_frames_huge": dict(num_ref=BIG),
    "crop_huge": dict(crop=(0, BIG, 0, 0)),
    "crop_eats_picture": dict(crop=(24, 24, 0, 0)),
}


@pytest.mark.parametrize("name", sorted(HOSTILE_SPS))
def test_hostile_sps_is_rejected_not_decoded(name):
    """A crafted ~10-byte SPS must not abort the process (std::length_error), shift by >= 32 or reach the device with a negative pitch:
    the parameter set is refused, the slice that refers to it fails cleanly and no frame comes out."""
    data = _sps(**HOSTILE_SPS[name]) + _pps() + _idr_slice()
    with api.JmAmdDec(0, 1, options={"parse_only": 1}) as d:
        n = d.decode_stream(data, keep=False)
        assert n == 0
        assert d.stat("errors") >= 1
        assert d.stat("coded_width") == 0 and d.stat("coded_height") == 0


@pytest.mark.parametrize("kw", [dict(nref0=BIG), dict(nref0=32), dict(init_qp=40), dict(cqo=13), dict(cqo=-13)])
def test_hostile_pps_is_rejected(kw):
    data = _sps() + _pps(**kw) + _idr_slice()
    with api.JmAmdDec(0, 1, options={"parse_only": 1}) as d:
        assert d.decode_stream(data, keep=False) == 0
        assert d.stat("errors") >= 1


def test_well_formed_handwritten_sps_is_accepted():
    """The same bit writer with in-range values activates a 96x80 sequence (so the rejections above are due to the values, not the writer)."""
    data = _sps() + _pps() + _idr_slice()
    with api.JmAmdDec(0, 1, options={"parse_only": 1}) as d:
        d.decode_stream(data, keep=False)
        assert (d.stat("coded_width"), d.stat("coded_height")) == (96, 80)


def test_crop_only_change_at_idr_updates_the_display_size(oracle):
    """a new SPS that only changes the cropping starts a new sequence as far as frame sizes go."""
    a = streams.generate(width=96, height=80, frames=2, gop=2, seed=5)
    b = streams.generate(width=90, height=70, frames=2, gop=2, seed=6)       # same 6x5 macroblocks, cropped to 90x70
    sizes = []
    with api.JmAmdDec(0, 1, options={"parse_only": 1}) as d:
        for nal in api.split_nalus(a + b):
            _, got = api.jm_nvdec_decode_frame(nal, len(nal), d.h)
            if got == 1:
                sizes.append(api.jm_nvdec_stream_info(d.h)); d._pull(None)
        while not api.jm_nvdec_is_exit(d.h):
            _, got = api.jm_nvdec_decode_frame(None, 0, d.h)
            if got == 1:
                sizes.append(api.jm_nvdec_stream_info(d.h)); d._pull(None)
    assert sizes == [(96, 80)] * 2 + [(90, 70)] * 2


def _job_digest(data, fast):
    with api.JmAmdDec(0, 1, options={"parse_only": 1, "job_digest": 1, "fast_parse": 1 if fast else 0}) as d:
        n = d.decode_stream(data, keep=False)
        return d.stat("job_digest") & (2 ** 64 - 1), n, d.stat("errors"), d.stat("job_bytes")


@pytest.mark.parametrize("name", sorted(PARITY_CASES))
def test_fast_p_slice_path_builds_the_same_job_lists(name):
    """P_Skip / P_L0_16x16 macroblocks of CAVLC P slices take a shorter route through the parser (h264_cavlc.cpp, `fast`); the syntax digest
    above always runs the general route, so the two are compared here on what the device gets: records, coefficients, vectors."""
    data = streams.generate(**PARITY_CASES[name])
    a, b = _job_digest(data, True), _job_digest(data, False)
    assert a == b and a[2] == 0 and a[1] == PARITY_CASES[name]["frames"]


@pytest.mark.parametrize("seed", range(36))
def test_fast_p_slice_path_on_random_tool_mixes(seed):
    """CAVLC and CABAC, with and without the 8x8 transform, B pictures (whose direct prediction reads the motion the fast path leaves behind), weighted
    prediction (takes the general path), constrained intra prediction, several slices and references."""
    rng = random.Random(900 + seed)
    cabac = rng.choice([0, 0, 1])
    bframes = rng.choice([0, 0, 2]) if seed % 3 else 0
    kw = dict(width=rng.choice([48, 96, 176, 320]), height=rng.choice([48, 80, 144, 240]), frames=rng.choice([4, 7, 10]), gop=rng.choice([3, 5, 30]),
              qp=rng.choice([20, 26, 32, 40]), mode=rng.choice([0, 1]), num_ref=rng.choice([1, 2, 4]), slices=rng.choice([1, 2, 3]), seed=0x5000 + seed,
              deblock=rng.choice([0, 1, 2]), cabac=cabac, cabac_idc=rng.choice([0, 1, 2]) if cabac else 0, t8x8=rng.choice([0, 1]) if (cabac or bframes) else 0,
              cip=rng.choice([0, 0, 1]), bframes=bframes, poc_type=0 if bframes else 2, direct_temporal=rng.choice([0, 1]) if bframes else 0,
              wp=rng.choice([0, 0, 1, 2]) if (cabac or bframes) else 0)
    if bframes:
        kw["num_ref"] = max(2, kw["num_ref"]); kw["frames"] = 1 + 3 * rng.choice([1, 2, 3]); kw["gop"] = 30
    data = streams.generate(**kw)
    a, b = _job_digest(data, True), _job_digest(data, False)
    assert a == b and a[2] == 0 and a[1] == kw["frames"], kw


def test_fast_p_slice_path_full_size():
    data = streams.generate(**streams.config_c1(stream_id=3, frames=6))
    a, b = _job_digest(data, True), _job_digest(data, False)
    assert a == b and a[2] == 0 and a[1] == 6


@pytest.mark.parametrize("delay", [0, 2, 6])
def test_display_delay_holds_frames_back_but_loses_none(delay):
    """set_option("display_delay", n) mirrors the reference's ulMaxDisplayDelay (nv_dec.cpp:341): while input keeps coming a frame is handed out only
    when n pictures are still on their way; the end of the stream drains everything, in the same display order."""
    data = streams.generate(width=96, height=80, frames=12, gop=6, num_ref=2, seed=0x4D61, poc_type=0, nonref_period=3)
    with api.JmAmdDec(0, 1, options={"parse_only": 1, "display_delay": delay}) as d:
        n = d.decode_stream(data, keep=False)
        pocs = [d.stat(f"display_poc:{i}") for i in range(n)]
    with api.JmAmdDec(0, 1, options={"parse_only": 1}) as d0:
        n0 = d0.decode_stream(data, keep=False)
        pocs0 = [d0.stat(f"display_poc:{i}") for i in range(n0)]
    assert n == n0 == 12 and pocs == pocs0


@pytest.mark.parametrize("kw", [dict(width=352, height=288, frames=8, gop=4, qp=10, seed=0x4D92, num_ref=2),
                                dict(width=320, height=240, frames=9, gop=9, qp=12, seed=0x4D93, cabac=1, t8x8=1, bframes=2, poc_type=0, wp=1, mode=1)])
def test_job_slots_grow_on_demand_and_build_the_same_job_lists(kw, monkeypatch):
    """Job slots start at what an ordinary picture needs and grow when a picture does not fit (it is parsed again into a bigger slot; motion records and
    weight tables that do not fit behind the levels move the slot's contents): low-QP streams outgrow the starting size several times over.  The job
    lists the device would get -- and the syntax digest, which must not see the abandoned attempts -- equal those of worst-case slots."""
    data = streams.generate(**kw)

    def run():
        with api.JmAmdDec(0, 1, options={"parse_only": 1, "job_digest": 1, "digest": 1}) as d:
            n = len(d.decode_stream(data))
            return (d.stat("job_digest") & (2 ** 64 - 1), d.stat("syntax_digest") & (2 ** 64 - 1), n, d.stat("errors"),
                d.stat("job_bytes")), d.stat("job_regrown"), d.stat("job_slot_bytes")
    small, regrown, slot_bytes = run()
    monkeypatch.setenv("JM_AMD_DEC_JOB_WORST_CASE", "1")
    worst, regrown_w, slot_bytes_w = run()
    assert small == worst and small[3] == 0 and small[2] == kw["frames"]
    assert regrown >= 1 and regrown_w == 0
    assert slot_bytes < slot_bytes_w


def test_parse_pools_per_numa_node_in_the_many_gpus_one_process_mode(oracle, monkeypatch):
    """The drop-in mode 'one process, handles round robin over every GPU': handles of GPUs on different NUMA nodes use different parse pools (workers on the
    node's CPUs, page-locked job buffers from its memory).  No GPU here: JM_AMD_DEC_FAKE_NUMA places two parse-only 'devices' on two nodes; both handles
    decode concurrently through their own pool and build the oracle's syntax."""
    import threading
    monkeypatch.setenv("JM_AMD_DEC_FAKE_NUMA", "0:0,1:1")
    monkeypatch.delenv("JM_AMD_DEC_DEVICE", raising=False)
    data = streams.generate(width=176, height=144, frames=12, gop=6, mode=1, num_ref=2, seed=0x4D94, cabac=1)
    want = oracle.syntax_digest(data)
    got = {}

    def run(dev):
        with api.JmAmdDec(0, 1, options={"device": dev, "parse_only": 1, "digest": 1}) as d:
            n = len(d.decode_stream(data))
            got[dev] = (d.stat("syntax_digest") & (2 ** 64 - 1), d.stat("digest_mbs"), n, d.stat("errors"), d.stat("numa_node"), d.stat("threads"))
    ts = [threading.Thread(target=run, args=(dev,)) for dev in (0, 1)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for dev in (0, 1):
        assert got[dev][:4] == (want[0], want[1], 12, 0) and got[dev][4] == dev and got[dev][5] >= 1


def test_all_intra_stream_is_not_held_to_the_lent_worst_case_buffers(monkeypatch):
    """every picture of an intra-only stream is an I picture.  Three worst-case job buffers per handle are lent to I pictures; a
    picture that finds none free must not wait for one (the pipeline would run three deep and block the feeding thread) -- it takes an ordinary slot,
    which grows to I-picture size once and stays that size.  Fed in one chunk so that many pictures are in flight at once: every frame comes out, no
    error, the job bytes equal those of worst-case slots, and the slots grew a bounded number of times (not once per picture)."""
    data = streams.generate(width=352, height=288, frames=60, gop=1, qp=14, seed=0x4D95)

    def run():
        with api.JmAmdDec(0, 1, options={"parse_only": 1}) as d:
            n = len(d.decode_stream(None, chunks=[data]))
            return n, d.stat("errors"), d.stat("job_bytes"), d.stat("job_regrown"), d.stat("i_pictures")
    n, errors, job_bytes, regrown, n_i = run()
    monkeypatch.setenv("JM_AMD_DEC_JOB_WORST_CASE", "1")
    n_w, errors_w, job_bytes_w, regrown_w, _ = run()
    assert (n, errors, job_bytes) == (60, 0, job_bytes_w) and (n_w, errors_w, regrown_w) == (60, 0, 0) and n_i == 60
    assert 1 <= regrown <= 2 * 24, regrown          # at most a grow event or two per job slot (24 slots), whatever the number of pictures
