"""HEVC on the GPU: the HIP path behind the jm_nvdec_* C ABI (codec_type 1) against the CPU oracle, bit-exact.

Small seeded streams for every tool the generator can produce, the committed golden vectors, and -- at BASELINE.json's sizes
(1080p, 4K) -- a few frames of the C3 configuration (64x64 CTBs, SAO + deblocking, random-access GOP 8)."""
import json
import os
import threading

import pytest

import jmcodec_amd
from tools import streams
from test_hevc_oracle import HEVC_CASES
from util import GOLDEN, md5

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    return streams.OracleHevc()


def gpu_decode(data, out_fmt=1, options=None):
    with jmcodec_amd.JmAmdDec(1, out_fmt, options=dict({"device": 0}, **(options or {}))) as d:
        frames = d.decode_stream(data)
        return frames, d.stat("errors")


@pytest.mark.parametrize("name", sorted(HEVC_CASES))
def test_bit_exact_vs_oracle(oracle, name):
    data = streams.generate_hevc(**HEVC_CASES[name])
    want, n, w, h = oracle.decode(data, 1)
    frames, errors = gpu_decode(data)
    assert errors == 0 and len(frames) == n
    fs = w * h * 3 // 2
    for i, f in enumerate(frames):
        assert f == want[i * fs:(i + 1) * fs], f"{name}: frame {i} differs from the oracle"


def test_golden_vectors_i420_and_nv12():
    meta = json.load(open(os.path.join(GOLDEN, "golden_hevc.json")))
    for name, m in meta.items():
        data = open(os.path.join(GOLDEN, name + ".h265"), "rb").read()
        for fmt, key in ((1, "md5_i420"), (0, "md5_nv12")):
            frames, errors = gpu_decode(data, fmt)
            assert errors == 0 and len(frames) == m["frames"]
            assert md5(b"".join(frames)) == m[key], (name, key)


def test_c3_configuration_1080p_and_4k(oracle):
    for w, h, frames in ((1920, 1080, 9), (3840, 2160, 3)):
        data = streams.generate_hevc(**streams.config_c3(frames=frames, width=w, height=h))
        want, n, ww, hh = oracle.decode(data, 1)
        got, errors = gpu_decode(data)
        assert errors == 0 and len(got) == n == frames and (ww, hh) == (w, h)
        assert b"".join(got) == want, f"{w}x{h}"


def test_concurrent_hevc_and_h264_streams(oracle):
    """independent handles of both codecs on one device (the per-device engine batches them in separate lanes)"""
    o264 = streams.Oracle()
    jobs = []
    for i in range(6):
        if i % 2:
            data = streams.generate(width=96, height=80, frames=8, gop=4, seed=100 + i, cabac=1, bframes=1, num_ref=2, poc_type=0)
            jobs.append((0, data, o264.decode(data, 1)[0]))
        else:
            data = streams.generate_hevc(**dict(HEVC_CASES["b_gop2"], seed=200 + i))
            jobs.append((1, data, oracle.decode(data, 1)[0]))
    results = [None] * len(jobs)

    def run(k):
        codec, data, _ = jobs[k]
        with jmcodec_amd.JmAmdDec(codec, 1, options={"device": 0}) as d:
            results[k] = b"".join(d.decode_stream(data))
    ts = [threading.Thread(target=run, args=(k,)) for k in range(len(jobs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for k, (_, _, want) in enumerate(jobs):
        assert results[k] == want, f"stream {k}"


def test_corrupt_hevc_streams_do_not_hang():
    import numpy as np
    data = streams.generate_hevc(**HEVC_CASES["b_gop2"])
    rng = np.random.default_rng(21)
    for trial in range(12):
        b = bytearray(data)
        for p in rng.integers(60, len(b), size=2):
            b[p] ^= 1 << int(rng.integers(0, 8))
        with jmcodec_amd.JmAmdDec(1, 1, options={"device": 0}) as d:
            d.decode_stream(bytes(b), keep=False)


def test_hevc_through_the_intel_push_pull_api_and_hvcc(oracle):
    """jm_intel_dec_init(1, ...) (codec enum jm_intel_dec.h:33) over the same engine, fed in arbitrary chunks; and hvcC + length-prefixed packets"""
    from jmcodec_amd import api
    data = streams.generate_hevc(**HEVC_CASES["b_gop8"])
    want, n, w, h = oracle.decode(data, 1)
    got, info, sinfo, biggest = api.intel_push_pull(data, codec_type=1, max_push=1500)      # the reference loop in pushes of at most 1,500 bytes
    assert biggest == 1500 and sinfo[:3] == (0, w, h)
    assert len(got) == n and b"".join(got) == want
    from jmcodec_amd import api
    rec, packets = api.annexb_to_hvcc(data)
    with jmcodec_amd.JmAmdDec(1, 1, options={"device": 0}, extra_data=rec) as d:
        frames = d.decode_stream(None, chunks=packets)
    assert b"".join(frames) == want


def test_cra_start_and_end_of_sequence_on_gpu(oracle):
    from test_hevc_oracle import cut_at_second_irap
    data = streams.generate_hevc(**HEVC_CASES["open_gop"])
    head, tail = cut_at_second_irap(data)
    for s in (tail, head + b"\x00\x00\x01\x48\x01" + tail):
        want, n, _, _ = oracle.decode(s, 1)
        frames, errors = gpu_decode(s)
        assert errors == 0 and len(frames) == n and b"".join(frames) == want


def test_resolution_change_between_sequences(oracle):
    a = streams.generate_hevc(**HEVC_CASES["p_real"])
    b = streams.generate_hevc(**dict(HEVC_CASES["crop_ctb32"], seed=77))
    data = a + b + a
    with jmcodec_amd.JmAmdDec(1, 1, options={"device": 0}) as d:
        frames = d.decode_stream(data)
        assert d.stat("errors") == 0
    # the oracle's convenience call reports one size: compare per sequence
    want = b"".join(oracle.decode(s, 1)[0] for s in (a, b, a))
    assert b"".join(frames) == want


def test_c3_4k_whole_gop8_pyramid():
    """BASELINE config C3 at full size with a COMPLETE random-access GOP-8 pyramid: 3840x2160 HEVC Main, 64x64 CTB, SAO + deblocking,
    17 frames (I, two full B pyramids of depth 3), every frame compared with the HEVC CPU oracle."""
    data = streams.generate_hevc(**streams.config_c3(frames=17))
    want, n, w, h = streams.OracleHevc().decode(data, 1)
    assert (w, h, n) == (3840, 2160, 17)
    fs = w * h * 3 // 2
    with jmcodec_amd.JmAmdDec(1, 1) as d:
        frames = d.decode_stream(data)
        assert d.stat("errors") == 0
        assert d.stat("b_pictures") >= 12
    assert len(frames) == 17
    for i, f in enumerate(frames):
        assert f == want[i * fs:(i + 1) * fs], f"frame {i} differs from the oracle"


def test_c3_full_length_4k_hevc():
    """BASELINE config C3 at its full size AND length (120 frames): 3840x2160 HEVC Main, 64x64 CTB, SAO + deblocking, random-access GOP 8,
    IDR every 32 -- four IDR periods (generated as four closed periods on four threads: a 4K picture costs the generator seconds, and an IDR picture
    restarts everything anyway).  The first period (32 frames, four complete B pyramids) AND the last one (frames 96..119: surfaces, job slots and
    collocated-motion fields recycled many times by then) are compared with the CPU oracle frame by frame; the whole run through size-independent
    properties: 120 frames in display order, no errors, every frame distinct, and a second decode fed in 64 KiB chunks instead of NAL-per-call gives
    the same 120 digests (chunking invariance + run-to-run determinism with the pictures of a stream parsed concurrently)."""
    from concurrent.futures import ThreadPoolExecutor
    from test_gpu_parity import _decode_digests, _md5_frames
    lengths = (32, 32, 32, 24)
    with ThreadPoolExecutor(4) as ex:                                     # generator and oracle are C libraries behind ctypes: the GIL is released
        periods = list(ex.map(lambda k: streams.generate_hevc(**streams.config_c3(frames=lengths[k], stream_id=k)), range(4)))
        data = b"".join(periods)
        assert [streams.idr_period(data, k, True) is not None for k in range(5)] == [True] * 4 + [False]
        wants = {k: ex.submit(lambda k=k: streams.OracleHevc().decode(periods[k], 1)) for k in (0, 3)}
        digs, kinds = _decode_digests(data, 3840, 2160, codec=1)
        # 4 IDR pictures; of the others about two in three are B pictures (the anchors of the GOP-8 pyramids are P)
        assert len(digs) == 120 and len(set(digs)) == 120 and sum(kinds) == 120 and kinds[0] == 4 and kinds[2] >= 70
        digs2, _ = _decode_digests(data, 3840, 2160, codec=1, chunks=[data[i:i + 65536] for i in range(0, len(data), 65536)])
        assert digs2 == digs
        for k, first in ((0, 0), (3, 96)):
            want, n, w, h = wants[k].result()
            assert (w, h, n) == (3840, 2160, lengths[k])
            assert digs[first:first + n] == _md5_frames(want, w * h * 3 // 2), f"IDR period {k} differs from the oracle"


def test_independent_pictures_of_a_stream_share_a_launch(oracle):
    """Round 3: the engine puts INDEPENDENT pictures of one HEVC handle -- the B pictures of one level of a random-access pyramid -- into one batch
    (each with its own pre-SAO work surface and residual scratch).  Eight handles decode GOP-8 pyramids concurrently, every frame of every handle is
    compared with the oracle, twice over: a picture that decodes into a surface the previous batch still packs out, or shares a work surface with a
    batch mate, shows up as a wrong frame here (the first form of the change corrupted a displayed frame at 16 streams)."""
    datas = [streams.generate_hevc(**streams.config_c3(frames=25, width=640, height=368, stream_id=40 + i)) for i in range(8)]
    wants = [oracle.decode(d, 1) for d in datas]
    for rep in range(2):
        got, errs, ppb = [None] * 8, [None] * 8, [None] * 8

        def run(i):
            with jmcodec_amd.JmAmdDec(1, 1) as dec:
                b0 = (dec.stat("eng_batches"), dec.stat("eng_batch_pics"))
                got[i] = dec.decode_stream(datas[i] * 2)                     # two IDR periods back to back
                errs[i] = dec.stat("errors")
        ts = [threading.Thread(target=run, args=(i,)) for i in range(8)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        for i in range(8):
            want, n, w, h = wants[i]
            fs = w * h * 3 // 2
            assert errs[i] == 0 and len(got[i]) == 2 * n
            for k, f in enumerate(got[i]):
                assert f == want[(k % n) * fs:(k % n + 1) * fs], f"pass {rep}, handle {i}: frame {k} differs from the oracle"


@pytest.mark.gpu
def test_mixed_codecs_concurrently_all_engine_lanes():
    """H.264 Baseline, H.264 High with B pictures and HEVC handles decoding at the same time on one device: ordinary lane (chain launches and stage kernels),
    H.264 intra lane, HEVC lane and HEVC I-picture lane all busy, 18 handles on 18 threads, every frame compared with the oracles."""
    import threading
    from util import ALL_CASES
    o4, oh = streams.Oracle(), streams.OracleHevc()
    jobs = []
    for rep in range(2):
        for name in ("real_qvga", "fuzz_multiref_slices", "b_real_spatial", "b_fuzz_cabac_high", "high_real_qvga", "fmo0_cabac_b_crop"):
            d = streams.generate(**dict(ALL_CASES[name], seed=ALL_CASES[name].get("seed", 1) + 1000 * rep))
            jobs.append((0, name, d, o4.decode(d, 1)[0]))
        for kw in (dict(width=320, height=240, frames=9, gop=8, num_ref=2, seed=0x4D70 + rep, sdh=1), dict(width=176, height=144, frames=17, gop=8, num_ref=3,
            seed=0x4D80 + rep, mode=1),
                   dict(width=352, height=288, frames=6, gop=0, num_ref=1, seed=0x4D90 + rep)):
            d = streams.generate_hevc(**kw)
            jobs.append((1, "hevc %dx%d" % (kw["width"], kw["height"]), d, oh.decode(d, 1)[0]))
    got, errs = [None] * len(jobs), [None] * len(jobs)

    def run(i):
        codec, _, d, _ = jobs[i]
        with jmcodec_amd.JmAmdDec(codec, 1) as dec:
            got[i] = b"".join(dec.decode_stream(d))
            errs[i] = dec.stat("errors")
    ts = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i, (codec, name, _, want) in enumerate(jobs):
        assert errs[i] == 0 and got[i] == want, (codec, name)
