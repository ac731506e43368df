"""Parity tests proper (-m gpu): the HIP decode path, driven through the C ABI exactly like the reference harness
drives jm_nvdec_* (test_nv_dec.cpp:163-259), must be BIT-EXACT against the CPU oracle / the committed golden vectors."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import jmcodec_amd
from jmcodec_amd import api
from tools import streams
from util import ALL_CASES as PARITY_CASES, golden_meta, golden_stream, md5

pytestmark = pytest.mark.gpu


def gpu_decode(data, out_fmt=1, **opts):
    with api.JmAmdDec(0, out_fmt, options=opts) as d:
        frames = d.decode_stream(data)
        assert d.stat("errors") == 0
        return frames


def first_diff(a, b, w, h):
    a = np.frombuffer(a, np.uint8); b = np.frombuffer(b, np.uint8)
    idx = np.flatnonzero(a != b)
    if not len(idx):
        return "identical"
    k = int(idx[0])
    return f"{len(idx)} bytes differ, first at byte {k} ({'luma x=%d y=%d' % (k % w, k // w) if k < w * h else 'chroma'}) got {a[k]} want {b[k]}"


def test_device_present_and_extension_loaded():
    assert jmcodec_amd.jm_nvdec_is_hw_support()
    assert any("libjm_amd_dec.so" in l for l in open("/proc/self/maps"))


@pytest.mark.parametrize("name", sorted(PARITY_CASES))
def test_bit_exact_vs_oracle(oracle, name):
    data = streams.generate(**PARITY_CASES[name])
    want, n, w, h = oracle.decode(data, 1)
    frames = gpu_decode(data)
    assert len(frames) == n
    fs = w * h * 3 // 2
    for i, f in enumerate(frames):
        assert f == want[i * fs:(i + 1) * fs], f"{name} frame {i}: " + first_diff(f, want[i * fs:(i + 1) * fs], w, h)


@pytest.mark.parametrize("name", sorted(golden_meta()))
@pytest.mark.parametrize("fmt", [0, 1])
def test_golden_vectors(name, fmt):
    m = golden_meta()[name]
    frames = gpu_decode(golden_stream(name), out_fmt=fmt)
    assert len(frames) == m["frames"]
    assert md5(b"".join(frames)) == m["md5_nv12" if fmt == 0 else "md5_i420"]
    if fmt == 1:
        assert [md5(f) for f in frames] == m["md5_frames_i420"]


def test_full_size_1080p_baseline(oracle):
    """BASELINE config 1 at full size (1920x1080, coded 1920x1088, crop_bottom 4), one GOP prefix."""
    data = streams.generate(**streams.config_c1(frames=8))
    want, n, w, h = oracle.decode(data, 1)
    assert (w, h, n) == (1920, 1080, 8)
    frames = gpu_decode(data)
    fs = w * h * 3 // 2
    assert len(frames) == n
    for i, f in enumerate(frames):
        assert f == want[i * fs:(i + 1) * fs], f"frame {i}: " + first_diff(f, want[i * fs:(i + 1) * fs], w, h)


def test_full_size_1080i_field_pictures(oracle):
    """Interlace at full size: 1920x1080 in a 1920x1088 surface (34 macroblock rows per field), Main profile CABAC, every picture a frame or two field
    pictures; three deblocking bands per field."""
    data = streams.generate(width=1920, height=1080, frames=8, gop=8, seed=301, paff=1, num_ref=2, cabac=1, qp=30, search=8)
    want, n, w, h = oracle.decode(data, 1)
    assert (w, h, n) == (1920, 1080, 8)
    tools = oracle.tools(data)
    assert tools.get("field-pictures", 0) >= 4 and tools.get("cross-parity-blocks", 0) > 0
    frames = gpu_decode(data)
    fs = w * h * 3 // 2
    assert len(frames) == n
    for i, f in enumerate(frames):
        assert f == want[i * fs:(i + 1) * fs], f"frame {i}: " + first_diff(f, want[i * fs:(i + 1) * fs], w, h)


def test_a_field_without_partner_on_device(oracle):
    """The second field of the last frame never arrives: the frame is shown with the lines of the decoded field repeated (k_packout lone_field), as the
    oracle shows it; and a stream that goes on with a FRAME after a lone field keeps decoding."""
    kw = dict(width=96, height=64, frames=4, gop=4, seed=302, paff=2, num_ref=2)
    data = streams.generate(**kw)
    starts = [i for i in range(len(data) - 4) if data[i:i + 4] == b"\0\0\0\1" or (data[i:i + 3] == b"\0\0\1" and data[i - 1:i] != b"\0")]
    cut = data[:starts[-1]]
    want, n, w, h = oracle.decode(cut, 1)
    assert n == 4 and oracle.tools(cut).get("lone-fields",
        0) == 0          # (the oracle counts a lone field when the NEXT picture starts; here the stream ends)
    assert b"".join(gpu_decode(cut)) == want
    # the same, followed by another coded video sequence: the lone field is completed when the IDR picture starts
    both = cut + streams.generate(**dict(kw, seed=303, paff=1))
    want2, n2, _, _ = oracle.decode(both, 1)
    assert n2 == 8 and oracle.tools(both).get("lone-fields", 0) == 1
    assert b"".join(gpu_decode(both)) == want2


def test_field_picture_streams_beside_progressive_ones(oracle):
    """Field pictures, frame pictures of interlace-capable streams and progressive streams in the same batches."""
    cases = ["paff_adaptive_fuzz", "real_qvga", "paff_fields_cabac", "fuzz_multiref_slices", "paff_poc1_t8x8_scaling", "paff_real_qvga"]
    datas = [streams.generate(**PARITY_CASES[c]) for c in cases]
    wants = [oracle.decode(d, 1)[0] for d in datas]
    got = [None] * len(cases)

    def run(i):
        got[i] = b"".join(gpu_decode(datas[i]))
    for _ in range(3):
        ts = [threading.Thread(target=run, args=(i,)) for i in range(len(cases))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        for i in range(len(cases)):
            assert got[i] == wants[i], cases[i]


def test_full_size_1080p_high_cabac(oracle):
    """1920x1080 High profile: CABAC, 8x8 transform, Intra8x8 (the tools of BASELINE config 2 that I/P streams use)."""
    kw = streams.config_c1(frames=5)
    kw.update(cabac=1, t8x8=1, qp=30, seed=0x4A4D0200)
    data = streams.generate(**kw)
    want, n, w, h = oracle.decode(data, 1)
    assert (w, h, n) == (1920, 1080, 5)
    frames = gpu_decode(data)
    fs = w * h * 3 // 2
    assert len(frames) == n
    for i, f in enumerate(frames):
        assert f == want[i * fs:(i + 1) * fs], f"frame {i}: " + first_diff(f, want[i * fs:(i + 1) * fs], w, h)


def test_full_size_4k_high_ibbp(oracle):
    """BASELINE config 2 at full size (3840x2160 High: CABAC, 8x8 transform, I B B P, two references), first pictures."""
    data = streams.generate(**streams.config_c2(frames=4))
    want, n, w, h = oracle.decode(data, 1)
    assert (w, h, n) == (3840, 2160, 4)
    with api.JmAmdDec(0, 1) as d:
        frames = d.decode_stream(data)
        assert d.stat("errors") == 0 and d.stat("b_pictures") == 2
    fs = w * h * 3 // 2
    assert len(frames) == n
    for i, f in enumerate(frames):
        assert f == want[i * fs:(i + 1) * fs], f"frame {i}: " + first_diff(f, want[i * fs:(i + 1) * fs], w, h)


def test_b_streams_concurrently(oracle):
    """Several B-picture streams at once: direct prediction waits for the colocated picture's motion across the worker pool."""
    from util import B_CASES
    names = sorted(B_CASES)[:6]
    datas = [streams.generate(**B_CASES[n]) for n in names]
    wants = [oracle.decode(d, 1)[0] for d in datas]
    got = [None] * len(names)

    def run(i):
        got[i] = b"".join(gpu_decode(datas[i]))
    for _ in range(3):
        ts = [threading.Thread(target=run, args=(i,)) for i in range(len(names))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        for i in range(len(names)):
            assert got[i] == wants[i], names[i]


def test_full_size_properties_300_frames():
    """Size-independent properties at BASELINE's full stream length where the scalar oracle would take too long:
    the same stream decoded twice, fed in different chunkings and by concurrent handles, gives identical frames;
    every IDR period of the looped stream decodes to the same bytes (idempotence of the closed GOP)."""
    base = streams.generate(**streams.config_c1(frames=30))
    data = base * 10                                    # 300 frames, 10 identical closed GOPs
    with api.JmAmdDec(0, 1) as d:
        digests = []
        count = [0]
        h = d.h
        buf = C.create_string_buffer(1920 * 1080 * 3 // 2)

        def pull():
            ret, n = api.jm_nvdec_output_frame(buf, len(buf), h)
            assert ret == n == len(buf)
            digests.append(md5(buf.raw)); count[0] += 1
        for nal in api.split_nalus(data):
            _, got = api.jm_nvdec_decode_frame(nal, len(nal), h)
            if got:
                pull()
        while not api.jm_nvdec_is_exit(h):
            _, got = api.jm_nvdec_decode_frame(None, 0, h)
            if got:
                pull()
        assert d.stat("errors") == 0
    assert count[0] == 300
    for g in range(1, 10):
        assert digests[30 * g:30 * g + 30] == digests[:30]
    assert len(set(digests[:30])) == 30                 # the content really moves


def test_chunking_invariance_on_device(oracle):
    data = golden_stream("ip_fuzz_96x80")
    want = b"".join(gpu_decode(data))
    with api.JmAmdDec(0, 1) as d:
        assert b"".join(d.decode_stream(data, chunks=[data])) == want
    with api.JmAmdDec(0, 1) as d:
        assert b"".join(d.decode_stream(data, chunks=[data[i:i + 97] for i in range(0, len(data), 97)])) == want
    with api.JmAmdDec(0, 1, options={"sync": 1}) as d:
        assert b"".join(d.decode_stream(data)) == want


def test_concurrent_handles(oracle):
    """N handles on N threads (config C4's per-GPU slice): same bytes as a lone handle."""
    cases = ["fuzz_multiref_slices", "real_qvga", "fuzz_cip_offsets", "fuzz_poc0_nonref_idc2"] * 2
    datas = [streams.generate(**PARITY_CASES[c]) for c in cases]
    wants = [oracle.decode(d, 1)[0] for d in datas]
    got = [None] * len(cases)

    def run(i):
        got[i] = b"".join(gpu_decode(datas[i]))
    ts = [threading.Thread(target=run, args=(i,)) for i in range(len(cases))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(len(cases)):
        assert got[i] == wants[i], cases[i]


def test_32_streams_with_intra_pictures_ahead_of_their_turn(oracle):
    """Round 6 (Engine::form): with many streams the engine launches an intra-only picture AHEAD of its stream's earlier pictures when nothing they touch is
    its surface, and a stream changes lane as soon as the device is done with its pictures on the other lane.  32 handles on 32 threads (the headline's
    shape) decode streams with an IDR picture every 12 frames, frames left on the device so that the engine -- not the link -- is what the streams wait
    for: every frame of every handle must equal the oracle's, in order; the engine's counter says whether pictures really ran ahead (it depends on how
    far the host gets ahead of the device; the test lowers the rule's threshold so that the path is taken, and asserts that it was)."""
    import util
    kinds = [dict(width=1280, height=720, frames=72, gop=12, seed=900 + k, num_ref=1 + k % 2) for k in range(4)]
    datas = [streams.generate(**kinds[i % 4]) for i in range(32)]
    wants = {}
    for k in range(4):
        wants[k] = oracle.decode(datas[k], 1)
    got, errs = [None] * 32, [0] * 32
    lib = api.lib()
    hip = C.CDLL("libamdhip64.so")

    def run(i, d):
        # device-resident output: the frames wait in device memory until they are fetched with a plain hipMemcpy
        frames = []
        h = d.h
        dev, ln, gotf = C.c_void_p(0), C.c_int(0), C.c_int(0)
        buf = C.create_string_buffer(1280 * 720 * 3 // 2)

        def pull():
            if lib.jm_amddec_output_frame_device(C.byref(dev), C.byref(ln), h) > 0:
                assert ln.value == len(buf)
                hip.hipMemcpy(buf, dev, ln.value, 2)
                frames.append(buf.raw[:ln.value])
        # the whole stream in ONE call (as test_player hands over whole packets, test_player.cpp:253): every picture is dispatched before a frame is fetched,
        # so the queues behind the engine run as deep as the handle's job slots allow
        lib.jm_amddec_decode_frame(C.cast(C.c_char_p(datas[i]), C.c_void_p), len(datas[i]), C.byref(gotf), h)
        if gotf.value == 1:
            pull()
        while not lib.jm_amddec_is_exit(h):
            if lib.jm_amddec_decode_frame(None, 0, C.byref(gotf), h) != 0:
                break
            if gotf.value == 1:
                pull()
        got[i] = b"".join(frames)
        errs[i] = d.stat("errors")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    decs = [api.JmAmdDec(0, 1, options={"device_output": 1}) for _ in range(32)]
    try:
        # (the rule waits for deep queues -- ten earlier pictures of the stream pending --, which these copying threads do not build up: let every picture run
        #  ahead that the hazards allow)
        lib.jm_amddec_set_option(decs[0].h, b"early_intra_ahead", 1)
        early0 = decs[0].stat("eng_early_intra")
        ts = [threading.Thread(target=run, args=(i, decs[i])) for i in range(32)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        early = decs[0].stat("eng_early_intra") - early0
        lib.jm_amddec_set_option(decs[0].h, b"early_intra_ahead", 10)
    finally:
        for d in decs:
            d.close()
    for i in range(32):
        assert errs[i] == 0 and got[i] == wants[i % 4][0], f"handle {i} (kind {i % 4})"
    util.SESSION_NOTES.append(f"32-stream test: {early} intra pictures ran ahead of their stream's turn (of {32 * 6})")
    assert early > 0, "no intra picture ran ahead of its turn: the path this test is for was not taken"


def test_job_slots_and_long_chains_one_stream(oracle):
    """Round 6: one stream's chain launches hold what its job slots let the parser run ahead (option "job_slots", default 40 up to 1080p) -- up to 16 pictures
    with one or two streams active.  Same frames whatever the slot count and the chain depth; with 56 slots the launches are longer than with 8."""
    data = streams.generate(width=640, height=368, frames=96, gop=48, seed=77, num_ref=2)
    want, n, w, h = oracle.decode(data, 1)
    ours = _wait_until_the_gpu_is_ours()
    per_launch = {}
    for slots in (8, 56):
        with api.JmAmdDec(0, 1, options={"job_slots": slots}) as d:
            api.lib().jm_amddec_set_option(d.h, b"chain_depth", 0)
            b0, p0 = d.stat("eng_chain_batches"), d.stat("eng_chain_pics")
            frames = d.decode_stream(None, chunks=[data])
            assert d.stat("errors") == 0 and b"".join(frames) == want, slots
            nb = d.stat("eng_chain_batches") - b0
            per_launch[slots] = (d.stat("eng_chain_pics") - p0) / max(nb, 1)
    if ours and all(per_launch.values()):
        assert per_launch[56] > per_launch[8], per_launch
        assert per_launch[56] > 8.0, per_launch              # longer than round 5's cap of 8 pictures per stream and launch


def test_api_protocol_on_device():
    data = golden_stream("ip_real_96x80")
    h = api.jm_nvdec_create_handle()
    assert api.jm_nvdec_init(0, 1, None, 0, h) == 0
    buf = C.create_string_buffer(96 * 80 * 3 // 2)
    assert api.jm_nvdec_output_frame(buf, len(buf), h)[0] == -1
    frames = []
    for nal in api.split_nalus(data):
        ret, got = api.jm_nvdec_decode_frame(nal, len(nal), h)
        assert ret == 0
        if got:
            assert api.jm_nvdec_output_frame(buf, 16, h)[0] == -2
            ret, n = api.jm_nvdec_output_frame(buf, len(buf), h)
            assert ret == n == len(buf)
            frames.append(buf.raw)
    while not api.jm_nvdec_is_exit(h):
        ret, got = api.jm_nvdec_decode_frame(None, 0, h)
        if got:
            assert api.jm_nvdec_output_frame(buf, len(buf), h)[0] == len(buf)
            frames.append(buf.raw)
    assert md5(b"".join(frames)) == golden_meta()["ip_real_96x80"]["md5_i420"]
    info = api.jm_nvdec_show_dec_info(h)
    assert "Codec:\t\tH.264" in info and "Display:\t96 x 80" in info and "Frame Count:\t6" in info
    api.jm_nvdec_deinit(h)


@pytest.mark.parametrize("w,h,pitch", [(16, 16, 128), (90, 70, 128), (1920, 1080, 1920), (3840, 2160, 3840)])
@pytest.mark.parametrize("fmt", [0, 1])
def test_packout_kernel_vs_oracle(oracle, w, h, pitch, fmt):
    """k_packout alone (jm_amddec_packout_device) against the restatement of nv_dec.cpp:782-820.
    Device memory comes straight from the HIP runtime the library is linked to (no torch in this process:
    torch bundles its own libamdhip64 and two HIP runtimes cannot share one process)."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    rng = np.random.default_rng(w + h + fmt)
    src = rng.integers(0, 256, size=pitch * h * 3 // 2, dtype=np.uint8)
    n_out = w * h * 3 // 2
    d_src, d_dst = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_src), src.size) == 0 and hip.hipMalloc(C.byref(d_dst), n_out) == 0
    try:
        assert hip.hipMemcpy(d_src, src.ctypes.data_as(C.c_void_p), src.size, 1) == 0          # hipMemcpyHostToDevice
        assert api.lib().jm_amddec_packout_device(d_src, pitch, w, h, fmt, d_dst, None) == 0
        assert hip.hipDeviceSynchronize() == 0
        out = np.zeros(n_out, np.uint8)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), d_dst, n_out, 2) == 0             # hipMemcpyDeviceToHost
    finally:
        hip.hipFree(d_src); hip.hipFree(d_dst)
    rc, want = oracle.packout(src.tobytes(), pitch, w, h, fmt)
    assert rc == n_out
    assert out.tobytes() == want


def test_4k_picture_size(oracle):
    """3840x2160 (240 x 135 macroblocks: 9 chained bands per plane in the deblock kernel, 5 wavefront slots in the intra kernel)."""
    data = streams.generate(width=3840, height=2160, frames=2, gop=2, seed=0x4A4D0200, level_idc=51)
    want, n, w, h = oracle.decode(data, 1)
    assert (n, w, h) == (2, 3840, 2160)
    frames = gpu_decode(data)
    fs = w * h * 3 // 2
    assert len(frames) == 2
    for i, f in enumerate(frames):
        assert f == want[i * fs:(i + 1) * fs], f"frame {i}: " + first_diff(f, want[i * fs:(i + 1) * fs], w, h)


def test_mixed_resolutions_share_batches(oracle):
    """Handles of different picture sizes run concurrently: the engine puts their pictures into the same launches."""
    kws = [dict(width=1920, height=1080, frames=6, gop=3, seed=1), dict(width=176, height=144, frames=12, gop=4, mode=1, num_ref=2, seed=2),
           dict(width=640, height=360, frames=10, gop=5, seed=3), dict(width=90, height=70, frames=12, gop=6, mode=1, seed=4),
           dict(width=1280, height=720, frames=8, gop=4, mode=1, slices=2, seed=5), dict(width=16, height=16, frames=20, gop=5, mode=1, seed=6)]
    datas = [streams.generate(**kw) for kw in kws]
    wants = [oracle.decode(d, 1)[0] for d in datas]
    got = [None] * len(kws)

    def run(i):
        got[i] = b"".join(gpu_decode(datas[i]))
    for _ in range(2):
        ts = [threading.Thread(target=run, args=(i,)) for i in range(len(kws))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        for i in range(len(kws)):
            assert got[i] == wants[i], kws[i]


def test_corrupt_streams_do_not_crash_or_hang():
    """Damaged input must never crash or dead-lock the pipeline (reference policy: errors are swallowed, nv_dec.cpp:394-402).
    Covers CAVLC, CABAC / 8x8 transform and B pictures (direct prediction waits on another picture's motion field)."""
    from util import B_CASES
    bases = [golden_stream("ip_fuzz_96x80"), golden_stream("high_cabac_fuzz_96x80"), streams.generate(**B_CASES["b_fuzz_cabac_high"]),
             streams.generate(**B_CASES["b_fuzz_temporal_noinf8"])]
    rng = np.random.default_rng(7)
    for base in bases:
        for trial in range(10):
            b = bytearray(base)
            for _ in range(1 + trial % 4):
                p = int(rng.integers(40, len(b)))
                b[p] ^= 1 << int(rng.integers(0, 8))
            if trial % 3 == 0:
                b = b[:int(rng.integers(100, len(b)))]
            with api.JmAmdDec(0, 1) as d:
                frames = d.decode_stream(bytes(b))
                assert len(frames) <= 12
                for f in frames:
                    assert len(f) == 96 * 80 * 3 // 2


def test_arbitrary_slice_order_on_device(oracle):
    """Baseline's arbitrary slice order: the slices of every picture rearranged; same frames as the oracle's (job lists are per macroblock, the kernels
    never see slice order)."""
    from test_oracle import reorder_slices
    data = streams.generate(width=176, height=144, frames=6, gop=6, mode=1, num_ref=2, slices=3, seed=78)
    want = oracle.decode(data, 1)[0]
    for order in ((2, 0, 1), (1, 2, 0)):
        assert b"".join(gpu_decode(reorder_slices(data, order))) == want, order


def test_corrupt_field_picture_streams_do_not_crash_or_hang():
    """The same for interlaced streams: damaged field pictures (lost second fields, parities that do not pair, broken marking operations, B fields whose
    colocated field never arrived) must neither crash nor dead-lock; every frame that comes out has the stream's size."""
    bases = [streams.generate(**PARITY_CASES[c]) for c in ("paff_adaptive_fuzz", "paff_adaptive_fuzz_cabac_wp", "paff_mixed_b_temporal_cabac",
        "paff_b_spatial")]
    sizes = [(PARITY_CASES[c]["width"], PARITY_CASES[c]["height"]) for c in ("paff_adaptive_fuzz", "paff_adaptive_fuzz_cabac_wp", "paff_mixed_b_temporal_cabac",
                                                                            "paff_b_spatial")]
    rng = np.random.default_rng(11)
    for base, (w, h) in zip(bases, sizes):
        for trial in range(12):
            b = bytearray(base)
            for _ in range(1 + trial % 4):
                p = int(rng.integers(40, len(b)))
                b[p] ^= 1 << int(rng.integers(0, 8))
            if trial % 3 == 0:
                b = b[:int(rng.integers(100, len(b)))]
            if trial % 4 == 1:                                  # drop one NAL unit out of the middle (often a whole field)
                st = [i for i in range(len(b) - 4) if b[i:i + 4] == b"\0\0\0\1" or (b[i:i + 3] == b"\0\0\1" and b[i - 1:i] != b"\0")]
                if len(st) > 6:
                    k = int(rng.integers(3, len(st) - 2))
                    b = b[:st[k]] + b[st[k + 1]:]
            with api.JmAmdDec(0, 1) as d:
                frames = d.decode_stream(bytes(b))
                assert len(frames) <= 16
                for f in frames:
                    assert len(f) == w * h * 3 // 2


def test_thirdparty_high_profile_stream(oracle):
    """High profile (CABAC, 8x8 transform, Intra8x8) on the device: bit-exact with the oracle on the third-party clip."""
    import json, os
    g = os.path.join(os.path.dirname(__file__), "golden")
    data = open(os.path.join(g, "thirdparty_realshort.h264"), "rb").read()
    m = json.load(open(os.path.join(g, "thirdparty.json")))["realshort"]
    want, n, w, h = oracle.decode(data, 1)
    frames = gpu_decode(data)
    fs = w * h * 3 // 2
    assert len(frames) == n == m["frames"]
    for i, f in enumerate(frames):
        assert f == want[i * fs:(i + 1) * fs], f"frame {i}: " + first_diff(f, want[i * fs:(i + 1) * fs], w, h)
    assert md5(b"".join(frames)) == m["md5_i420"]
    assert md5(b"".join(gpu_decode(data, out_fmt=0))) == m["md5_nv12"]


def test_intel_push_pull_api(oracle):
    """jm_intel_dec_* facade driven like test_intel_dec.cpp:78-102: push chunks while need_more_data, pull frames,
    set_eof, run until is_exit.  Frames must equal the oracle's, in display order; then the same through the YUV callback."""
    from util import B_CASES
    data = streams.generate(**B_CASES["b_fuzz_cabac_high"])
    want, n, w, h = oracle.decode(data, 1)
    for use_cb in (False, True):
        got, info, sinfo, biggest = api.intel_push_pull(data, callback=use_cb, max_push=777)      # arbitrary small chunks: NAL units cut anywhere
        assert biggest == 777 and sinfo[:3] == (0, w, h)
        assert "Frame Count:\t%d" % n in info
        assert len(got) == n
        assert b"".join(got) == want


def test_device_resident_output_and_argb(oracle):
    """with option device_output frames never cross PCIe; the device copy must hold the same bytes as the normal
    output, and the ARGB conversion (BT.601 limited range, integer) must match a numpy restatement of the same formula."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    L = api.lib()
    L.jm_amddec_output_frame_device.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]
    L.jm_amddec_output_argb_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    data = golden_stream("high_cabac_fuzz_96x80")
    for fmt in (1, 0):
        want, n, w, h = oracle.decode(data, fmt)
        fs = w * h * 3 // 2
        pitch = w * 4 + 64
        d_argb = C.c_void_p()
        assert hip.hipMalloc(C.byref(d_argb), pitch * h) == 0
        try:
            with api.JmAmdDec(0, fmt, options={"device_output": 1}) as d:
                count = 0
                for nal in api.split_nalus(data) + [None] * 64:
                    if api.jm_nvdec_is_exit(d.h):
                        break
                    _, got = api.jm_nvdec_decode_frame(nal, len(nal) if nal else 0, d.h)
                    if not got:
                        continue
                    dev, ln = C.c_void_p(), C.c_int(0)
                    assert L.jm_amddec_output_frame_device(C.byref(dev), C.byref(ln), d.h) == fs and ln.value == fs
                    host = np.zeros(fs, np.uint8)
                    assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), dev, fs, 2) == 0
                    ref = np.frombuffer(want, np.uint8, fs, count * fs)
                    assert np.array_equal(host, ref), f"fmt {fmt} frame {count}"
                    assert L.jm_amddec_output_argb_device(d_argb, pitch, d.h) == 0
                    argb = np.zeros(pitch * h, np.uint8)
                    assert hip.hipMemcpy(argb.ctypes.data_as(C.c_void_p), d_argb, pitch * h, 2) == 0
                    argb = argb.reshape(h, pitch)[:, :w * 4].reshape(h, w, 4).astype(np.int32)
                    Y = ref[:w * h].reshape(h, w).astype(np.int32)
                    if fmt == 1:
                        U = ref[w * h:w * h + w * h // 4].reshape(h // 2, w // 2); V = ref[w * h + w * h // 4:].reshape(h // 2, w // 2)
                    else:
                        uv = ref[w * h:].reshape(h // 2, w // 2, 2); U, V = uv[:, :, 0], uv[:, :, 1]
                    D = U.astype(np.int32).repeat(2, 0).repeat(2, 1) - 128; E = V.astype(np.int32).repeat(2, 0).repeat(2, 1) - 128
                    c = 298 * (Y - 16) + 128
                    R, G, B = np.clip((c + 409 * E) >> 8, 0, 255), np.clip((c - 100 * D - 208 * E) >> 8, 0, 255), np.clip((c + 516 * D) >> 8, 0, 255)
                    assert np.array_equal(argb[:, :, 0], B) and np.array_equal(argb[:, :, 1], G) and np.array_equal(argb[:, :, 2], R) and (argb[:, :,
                        3] == 255).all()
                    count += 1
                assert count == n
        finally:
            hip.hipFree(d_argb)


def _nv12_pitch_restated(frame, w, h, fmt, pitch):
    """numpy restatement of nv_enc.cpp:1022-1079 (cuMemcpy2D of the luma plane + InterleaveUV): tight I420 / NV12 -> pitch NV12 surface."""
    out = np.zeros((h + h // 2, pitch), np.uint8)
    out[:h, :w] = frame[:w * h].reshape(h, w)
    if fmt == 0:
        out[h:, :w] = frame[w * h:].reshape(h // 2, w)
    else:
        U = frame[w * h:w * h + w * h // 4].reshape(h // 2, w // 2); V = frame[w * h + w * h // 4:].reshape(h // 2, w // 2)
        out[h:, 0:w:2] = U; out[h:, 1:w:2] = V
    return out


@pytest.mark.parametrize("w,h,pitch", [(16, 16, 16), (90, 70, 131), (1920, 1080, 2048), (3840, 2160, 3840)])
@pytest.mark.parametrize("fmt", [1, 0])
def test_encoder_preprocessing_kernel_vs_restatement(w, h, pitch, fmt):
    """jm_amddec_i420_to_nv12_device against the numpy restatement, including a pitch that is not a multiple of 4."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]; hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L = api.lib()
    L.jm_amddec_i420_to_nv12_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(w * 31 + h + fmt)
    frame = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
    rows = h + h // 2
    d_src, d_dst = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_src), frame.size) == 0 and hip.hipMalloc(C.byref(d_dst), rows * pitch) == 0
    try:
        assert hip.hipMemcpy(d_src, frame.ctypes.data_as(C.c_void_p), frame.size, 1) == 0
        assert hip.hipMemset(d_dst, 0, rows * pitch) == 0
        assert L.jm_amddec_i420_to_nv12_device(d_src, w, h, fmt, d_dst, pitch, None) == 0
        got = np.zeros(rows * pitch, np.uint8)
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), d_dst, got.size, 2) == 0
        assert np.array_equal(got.reshape(rows, pitch), _nv12_pitch_restated(frame, w, h, fmt, pitch))    # padding bytes stay untouched (zero)
        assert L.jm_amddec_i420_to_nv12_device(d_src, w + 1, h, fmt, d_dst, pitch, None) == -1 and L.jm_amddec_i420_to_nv12_device(d_src, w, h, fmt, d_dst,
            w - 1, None) == -1
    finally:
        hip.hipFree(d_src); hip.hipFree(d_dst)


def test_decode_to_encoder_surface_on_device(oracle):
    """the decoder's current frame (I420 at init) as a pitch NV12 encoder surface, device to device; it must equal the
    oracle's NV12 frame of the same stream laid out at that pitch."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]; hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L = api.lib()
    L.jm_amddec_output_nv12_pitch_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    data = golden_stream("high_cabac_fuzz_96x80")
    want_nv12, n, w, h = oracle.decode(data, 0)
    fs, pitch, rows = w * h * 3 // 2, 256, h + h // 2
    d_surf = C.c_void_p()
    assert hip.hipMalloc(C.byref(d_surf), rows * pitch) == 0
    try:
        with api.JmAmdDec(0, 1, options={"device_output": 1}) as d:
            assert L.jm_amddec_output_nv12_pitch_device(d_surf, pitch, d.h) == -1          # no frame yet
            count = 0
            for nal in api.split_nalus(data) + [None] * 64:
                if api.jm_nvdec_is_exit(d.h):
                    break
                _, got = api.jm_nvdec_decode_frame(nal, len(nal) if nal else 0, d.h)
                if not got:
                    continue
                assert hip.hipMemset(d_surf, 0, rows * pitch) == 0
                assert L.jm_amddec_output_nv12_pitch_device(d_surf, pitch, d.h) == 0
                surf = np.zeros(rows * pitch, np.uint8)
                assert hip.hipMemcpy(surf.ctypes.data_as(C.c_void_p), d_surf, surf.size, 2) == 0
                ref = np.frombuffer(want_nv12, np.uint8, fs, count * fs)
                assert np.array_equal(surf.reshape(rows, pitch), _nv12_pitch_restated(ref, w, h, 0, pitch)), f"frame {count}"
                count += 1
            assert count == n
    finally:
        hip.hipFree(d_surf)


def test_constrained_intra_unavailable_samples_count_as_128(oracle):
    """Found by tools/gpu_sweep.py: with constrained_intra_pred the generator selected (then unconditionally, now with nc_corner=1) Intra4x4 Horizontal-Down
    next to an INTER corner neighbour (the sample is not available, so a conforming encoder would not); oracle, generator and the spin-wait kernel count
    such samples as 128 and the LDS intra wavefront has to as well.  The P pictures of this stream are dense enough in intra macroblocks to take the LDS
    wavefront (three deblocking / intra bands)."""
    kw = dict(width=320, height=520, frames=7, qp=18, gop=4, seed=153603, mode=1, deblock=0, num_ref=3, slices=1, cabac=1, cabac_idc=0, t8x8=1, bframes=2,
              direct_temporal=0, wp=1, dinf8=0, scaling=0, rplm=1, cip=1, chroma_qp_off=-4, alpha_off=3, beta_off=0, poc_type=0,
              nc_corner=1)            # the generator's explicit switch for this (non-conforming) choice
    data = streams.generate(**kw)
    want, n, w, h = oracle.decode(data, 1)
    with jmcodec_amd.JmAmdDec(0, 1) as d:
        frames = d.decode_stream(data)
        assert d.stat("errors") == 0
    assert (n, w, h) == (7, 320, 520) and b"".join(frames) == want


def test_resolution_change_between_sequences(oracle):
    """Three coded video sequences of different picture sizes and tool sets in one stream: surfaces, job rings and scratch are rebuilt
    at each new SPS, frames already decoded are still delivered, every frame is bit-exact."""
    a = streams.generate(width=96, height=80, frames=5, gop=5, mode=1, seed=1, num_ref=2)
    b = streams.generate(width=320, height=240, frames=4, gop=4, seed=2, cabac=1, t8x8=1)
    c = streams.generate(width=64, height=48, frames=6, gop=6, mode=1, seed=3, bframes=2)
    want = b"".join(oracle.decode(x, 1)[0] for x in (a, b, c))
    assert oracle.decode(a + b + c, 1)[0] == want
    for _ in range(2):
        frames = gpu_decode(a + b + c)
        assert [len(f) for f in frames] == [96 * 80 * 3 // 2] * 5 + [320 * 240 * 3 // 2] * 4 + [64 * 48 * 3 // 2] * 6
        assert b"".join(frames) == want


# ---- BASELINE configs at their real shape -----------------------------------------------------------
def _md5_frames(blob, fs):
    return [md5(blob[i:i + fs]) for i in range(0, len(blob), fs)]


@pytest.mark.parametrize("fetch", ["1/2", "0/1", "1/1", "direct"])
def test_c4_slice_8x1080p_concurrent(oracle, fetch, monkeypatch):
    """BASELINE config C4's per-GPU slice: 8 DISTINCT 1080p Baseline streams (SURVEY 8d seeds, stream_id 0..7) decoded concurrently by 8
    handles on 8 threads, one IDR period (30 frames) each, every frame compared with the CPU oracle.  JM_AMD_DEC_OUT_FETCH forces the
    output routes: alternating, all through pinned slots, all fetched from device staging, all by DMA into the caller's registered buffer."""
    from concurrent.futures import ThreadPoolExecutor
    monkeypatch.setenv("JM_AMD_DEC_OUT_FETCH", fetch)
    fs = 1920 * 1080 * 3 // 2
    with ThreadPoolExecutor(8) as ex:
        datas = list(ex.map(lambda sid: streams.generate(**streams.config_c1(stream_id=sid, frames=30)), range(8)))
        wants = list(ex.map(lambda d: _md5_frames(oracle.decode(d, 1)[0], fs), datas))
    assert len(set(datas)) == 8 and all(len(w) == 30 for w in wants)
    got, errs = [None] * 8, [None] * 8

    def run(i):
        with api.JmAmdDec(0, 1) as d:
            got[i] = [md5(f) for f in d.decode_stream(datas[i])]
            errs[i] = d.stat("errors")
            assert fetch != "direct" or d.stat("direct_frames") == 30
    ts = [threading.Thread(target=run, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(8):
        assert errs[i] == 0
        assert got[i] == wants[i], f"stream {i}: first differing frame {next(k for k in range(30) if k >= len(got[i]) or got[i][k] != wants[i][k])}"


def _decode_digests(data, w, h, codec=0, chunks=None):
    """Frame MD5s in display order, through the C ABI exactly like test_nv_dec (decode_frame / output_frame)."""
    digs = []
    buf = C.create_string_buffer(w * h * 3 // 2)
    with api.JmAmdDec(codec, 1) as d:
        def pull():
            ret, n = api.jm_nvdec_output_frame(buf, len(buf), d.h)
            assert ret == n == len(buf)
            digs.append(md5(buf.raw))
        for nal in (chunks if chunks is not None else api.split_nalus(data)):
            _, got = api.jm_nvdec_decode_frame(nal, len(nal), d.h)
            if got:
                pull()
        while not api.jm_nvdec_is_exit(d.h):
            _, got = api.jm_nvdec_decode_frame(None, 0, d.h)
            if got:
                pull()
        assert d.stat("errors") == 0
        kinds = (d.stat("i_pictures"), d.stat("p_pictures"), d.stat("b_pictures"))
    return digs, kinds


def test_c2_full_length_4k_high_ibbp(oracle):
    """BASELINE config C2 at its full size AND length: 3840x2160 High (CABAC, 8x8 transform), I B B P with two references, 120 frames
    (four IDR periods).  The first IDR period is compared with the CPU oracle frame by frame; the whole run through size-independent
    properties: 120 frames in display order, no errors, and a second decode fed in 64 KiB chunks instead of NAL-per-call gives the
    same 120 digests (chunking invariance + run-to-run determinism at full batch depth)."""
    data = streams.generate(**streams.config_c2(frames=120))
    digs, kinds = _decode_digests(data, 3840, 2160)
    assert len(digs) == 120 and sum(kinds) == 120 and kinds[0] == 4 and kinds[2] >= 70      # 4 IDR pictures, ~2/3 B pictures
    nal_starts = [i for i in range(len(data) - 4) if data[i:i + 4] == b"\x00\x00\x00\x01" and (data[i + 4] & 31) == 7]
    assert len(nal_starts) >= 2                                                              # an SPS before every IDR picture
    want, n, w, h = oracle.decode(data[:nal_starts[1]], 1)
    assert (w, h, n) == (3840, 2160, 30)
    assert digs[:30] == _md5_frames(want, w * h * 3 // 2)
    digs2, _ = _decode_digests(data, 3840, 2160, chunks=[data[i:i + 65536] for i in range(0, len(data), 65536)])
    assert digs2 == digs
    assert len(set(digs)) == 120


def test_avcc_on_device(oracle):
    """SURVEY 8f f2, H.264 half, on the GPU: parameter sets as an avcC record through jm_nvdec_init(extra_data), packets as
    length-prefixed NAL units (test_player.cpp:221-226 without the mp4toannexb filter) -- same frames as the Annex-B form."""
    kw = dict(width=320, height=240, frames=12, gop=6, seed=31, cabac=1, t8x8=1, bframes=2, num_ref=2, poc_type=0)
    data = streams.generate(**kw)
    want, n, w, h = oracle.decode(data, 1)
    assert n == 12
    for ls in (4, 2):
        rec, packets = api.annexb_to_avcc(data, ls)
        with api.JmAmdDec(0, 1, extra_data=rec) as d:
            frames = d.decode_stream(None, chunks=packets)
            assert d.stat("errors") == 0
        assert b"".join(frames) == want, f"length_size {ls}"


def test_native_harness_on_device(oracle, tmp_path):
    """tools/test_amd_dec -- the native counterpart of the reference's test_nv_dec main loop (test_nv_dec.cpp:98-268), bound to the
    drop-in jm_nvdec_* symbols -- RUNS on the GPU: the YUV file it writes equals the oracle's frames, the info block reports them."""
    import os
    import subprocess
    from util import ROOT
    data = streams.generate(width=320, height=240, frames=9, gop=9, seed=41, cabac=1, bframes=2, num_ref=2, poc_type=0)
    want, n, w, h = oracle.decode(data, 1)
    src, dst = tmp_path / "in.h264", tmp_path / "out.yuv"
    src.write_bytes(data)
    exe = os.path.join(ROOT, "tools", "_build", "test_amd_dec")
    r = subprocess.run([exe, str(src), str(dst), "--chunk", "4096"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert f"Frame Count:\t{n}\n" in r.stdout and f"Display:\t{w} x {h}\n" in r.stdout
    assert dst.read_bytes() == want
    r = subprocess.run([exe, str(src), "--fmt", "0", "--loops", "4"], capture_output=True, text=True, timeout=300)     # NV12, looped input
    assert r.returncode == 0 and f"Frame Count:\t{4 * n}\n" in r.stdout and "Pixel Format:\tNV12\n" in r.stdout


# ---- chain launches (chain.hip): consecutive pictures of one stream in one launch ----------------------------------------------------
CHAIN_CASES = {
    # several deblocking bands (more than 16 macroblock rows), several references, sub-8x8 partitions, vectors that point far down
    # (no_intra: the fuzz generator codes no intra macroblocks in P / B pictures: chains of the two-role kernel k_chain)
    "p_multiband_fuzz": dict(width=352, height=416, frames=14, gop=14, mode=1, num_ref=3, seed=201, no_intra=1),
    "p_multiband_real": dict(width=640, height=368, frames=12, gop=12, seed=202, search=12),
    "b_multiband_cabac": dict(width=352, height=288, frames=13, gop=13, mode=1, num_ref=2, bframes=2, cabac=1, seed=203, poc_type=0, no_intra=1),
    "b_temporal_t8x8": dict(width=352, height=288, frames=13, gop=13, mode=1, num_ref=3, bframes=3, cabac=1, t8x8=1, direct_temporal=1, seed=206, poc_type=0,
    no_intra=1),
    "wp_slices_nonref": dict(width=320, height=272, frames=12, gop=12, mode=1, num_ref=2, wp=1, slices=3, nonref_period=3, seed=204, poc_type=0, no_intra=1),
    "mmco_deblock_idc2": dict(width=320, height=272, frames=16, gop=16, mode=1, num_ref=3, mmco=1, deblock=2, slices=2, seed=205, no_intra=1),
    # intra macroblocks scattered through P / B pictures, and several IDR periods: those pictures join the chains through the intra role (k_chain_i)
    "mixed_intra_p": dict(width=320, height=272, frames=16, gop=16, mode=1, num_ref=2, seed=207),
    "mixed_intra_b_cabac_i8x8": dict(width=352, height=288, frames=14, gop=7, mode=1, num_ref=2, bframes=2, cabac=1, t8x8=1, seed=208, poc_type=0),
    "idr_every_4_real": dict(width=640, height=368, frames=13, gop=4, seed=209),
    "cip_mixed_intra": dict(width=320, height=272, frames=12, gop=12, mode=1, num_ref=2, cip=1, seed=210),
}


_GPU_OURS_SEEN = []       # every answer of _wait_until_the_gpu_is_ours() in this session (the closing test reads it)


def _wait_until_the_gpu_is_ours():
    """Chain launches are off while another process has compute queues on the GPU (engine.cpp).  The native-harness test runs such a process, and the
    driver keeps its queues listed for a moment after it has exited: wait until the engine sees the GPU unshared before asserting that chains form."""
    import time
    tiny = streams.generate(width=64, height=48, frames=2, gop=2)
    for _ in range(100):
        with api.JmAmdDec(0, 1) as d:
            api.lib().jm_amddec_set_option(d.h, b"chain_depth", 0)       # a chain option makes the next batch look again
            d.decode_stream(None, chunks=[tiny])
            if d.stat("eng_gpu_shared") == 0:
                _GPU_OURS_SEEN.append(True)
                return True
        time.sleep(0.2)
    _GPU_OURS_SEEN.append(False)
    return False            # another process keeps compute queues on this GPU: no chain launches will form (results must still be right)


def _chains_must_have_formed(ours, what):
    """a chain test may not pass vacuously.  The frames were compared already (against the stage kernels if no chain formed); whether the
    chain kernels themselves ran is an explicit outcome: asserted on a GPU the engine owns, a visible SKIP (-rs prints it) on a shared one."""
    if not ours:
        pytest.skip(f"GPU shared with another process for 20 s: no chain launch formed, {what} was compared on the stage kernels only")


@pytest.mark.parametrize("name", sorted(CHAIN_CASES))
def test_chain_launch_vs_oracle(oracle, name):
    """One stream decoded with chain launches of depth 1 (off), 3, 8 and 16 and with the tightest / a wide spacing of the pictures in the
    work list: every variant is bit-exact against the oracle, chains really formed, and no wait between workgroups timed out."""
    data = streams.generate(**CHAIN_CASES[name])
    want, n, w, h = oracle.decode(data, 1)
    ours = _wait_until_the_gpu_is_ours()
    for depth, lag in ((1, 24), (3, 20), (8, 24), (16, 64)):
        with api.JmAmdDec(0, 1) as d:
            lib = api.lib()
            assert lib.jm_amddec_set_option(d.h, b"chain_depth", depth) == 0 and lib.jm_amddec_set_option(d.h, b"chain_lag", lag) == 0
            before = d.stat("eng_chain_pics")
            # the whole stream in one call: every picture is pending when the engine forms its batches
            frames = d.decode_stream(None, chunks=[data])
            assert d.stat("errors") == 0 and d.stat("device_wait_errors") == 0
            chained = d.stat("eng_chain_pics") - before
            lib.jm_amddec_set_option(d.h, b"chain_depth", 0); lib.jm_amddec_set_option(d.h, b"chain_lag", 24)
        assert len(frames) == n
        assert b"".join(frames) == want, f"{name}: depth {depth} lag {lag} differs from the oracle"
        assert chained == 0 or depth > 1, (name, depth, chained)
        assert chained > 0 or depth == 1 or not ours, f"{name}: depth {depth}: no picture ran inside a chain launch on a GPU the engine owns"
    _chains_must_have_formed(ours, name)


def test_chain_launch_1080p_two_gops(oracle):
    """BASELINE config C1 through chain launches at full size: two IDR periods of one 1080p stream fed in one call (so that every P picture
    of an IDR period is pending at once), each frame compared with the oracle."""
    data = streams.generate(**streams.config_c1(stream_id=7, frames=40))
    want, n, w, h = oracle.decode(data, 1)
    fs = w * h * 3 // 2
    ours = _wait_until_the_gpu_is_ours()
    with api.JmAmdDec(0, 1) as d:
        before = d.stat("eng_chain_pics")
        frames = d.decode_stream(None, chunks=[data])
        chained = d.stat("eng_chain_pics") - before
        assert d.stat("errors") == 0
    assert len(frames) == n == 40
    for i, f in enumerate(frames):
        assert f == want[i * fs:(i + 1) * fs], f"frame {i}: " + first_diff(f, want[i * fs:(i + 1) * fs], w, h)
    assert chained >= 30 or not ours, f"only {chained} of 40 pictures ran inside chain launches on a GPU the engine owns"
    _chains_must_have_formed(ours, "the 1080p two-period stream")


def test_damaged_handover_is_reported_not_silent():
    """a wait between workgroups that gives up must surface as a decode error.  Debug option "debug_stall" makes the deblocking
    bands never publish their step counters, so the second band of a picture taller than 16 macroblock rows runs into its bounded wait:
    the handle reports errors > 0, device_wait_errors > 0 and says so in jm_amddec_last_error (stage kernels; chain launches off)."""
    data = streams.generate(width=176, height=288, frames=4, gop=4, seed=77)           # 18 macroblock rows: two bands
    lib = api.lib()
    with api.JmAmdDec(0, 1) as d:
        lib.jm_amddec_set_option(d.h, b"chain_depth", 1)
        lib.jm_amddec_set_option(d.h, b"debug_stall", 1)
        try:
            frames = d.decode_stream(None, chunks=[data])
            errs, werrs, msg = d.stat("errors"), d.stat("device_wait_errors"), lib.jm_amddec_last_error(d.h).decode()
        finally:
            lib.jm_amddec_set_option(d.h, b"debug_stall", 0)
            lib.jm_amddec_set_option(d.h, b"chain_depth", 0)
    assert len(frames) == 4                       # the pipeline still completes
    assert werrs > 0 and errs >= werrs
    assert "timed out" in msg
    with api.JmAmdDec(0, 1) as d:                 # and the engine is healthy afterwards
        assert b"".join(d.decode_stream(data)) == b"".join(gpu_decode(data))
        assert d.stat("errors") == 0


def test_kernel_without_two_list_code_says_so_when_it_meets_a_two_list_record():
    """k_recon_inter<HAS_BI = false> -- the instantiation for batches that hold no B / weighted picture -- must not decode a two-list record as a
    plain one if a future producer ever hands it one.  Debug option "debug_no_bi" launches that instantiation whatever the batch holds: on a stream with B
    pictures the handle reports errors, device_wait_errors and names the cause; without the option the same handle type decodes the stream cleanly."""
    data = streams.generate(width=320, height=240, frames=9, gop=9, seed=41, cabac=1, bframes=2, num_ref=2, poc_type=0)
    lib = api.lib()
    with api.JmAmdDec(0, 1) as d:
        lib.jm_amddec_set_option(d.h, b"chain_depth", 1)
        lib.jm_amddec_set_option(d.h, b"debug_no_bi", 1)
        try:
            frames = d.decode_stream(None, chunks=[data])
            errs, werrs, msg = d.stat("errors"), d.stat("device_wait_errors"), lib.jm_amddec_last_error(d.h).decode()
        finally:
            lib.jm_amddec_set_option(d.h, b"debug_no_bi", 0)
            lib.jm_amddec_set_option(d.h, b"chain_depth", 0)
    assert len(frames) == 9                       # the pipeline still completes
    assert werrs > 0 and errs >= werrs
    assert "compiled without" in msg, msg
    with api.JmAmdDec(0, 1) as d:                 # and the engine is healthy afterwards
        assert b"".join(d.decode_stream(data)) == b"".join(gpu_decode(data))
        assert d.stat("errors") == 0


def test_chain_launch_that_runs_out_of_time_is_decoded_again(oracle):
    """A chain launch assumes it can keep its bands resident; on a GPU it shares (another process, a long kernel of another stream) a wait inside it
    can run out of time.  That must cost time, not correctness: the engine decodes the launch's pictures -- and those of the lane's next batch, which
    read them -- again with the stage kernels.  "debug_stall" = 2 makes the chain launches of this stream fail that way (their bands never publish): the
    frames still equal the oracle's, no error is reported, and eng_chain_recoveries counts the event."""
    data = streams.generate(width=352, height=288, frames=12, gop=6, seed=78, num_ref=2)       # 18 macroblock rows, two IDR periods
    want, n, w, h = oracle.decode(data, 1)
    lib = api.lib()
    ours = _wait_until_the_gpu_is_ours()
    with api.JmAmdDec(0, 1) as d:
        lib.jm_amddec_set_option(d.h, b"chain_depth", 0)
        lib.jm_amddec_set_option(d.h, b"debug_stall", 2)           # 2 = chain launches only: the stage kernels that redo the pictures work
        before = d.stat("eng_chain_recoveries")
        try:
            frames = d.decode_stream(None, chunks=[data])
            errs, rec = d.stat("errors"), d.stat("eng_chain_recoveries") - before
        finally:
            lib.jm_amddec_set_option(d.h, b"debug_stall", 0)
            lib.jm_amddec_set_option(d.h, b"chain_depth", 0)     # (also ends the pause of chain launches that follows a recovery)
    assert len(frames) == n and b"".join(frames) == want and errs == 0
    assert rec >= 1 or not ours, "debug_stall 2 did not make a chain launch give up on a GPU the engine owns"
    _chains_must_have_formed(ours, "the recovery path")


@pytest.mark.parametrize("kw", [dict(num_ref=4, frames=48, gop=24), dict(num_ref=3, frames=39, gop=13, bframes=2, cabac=1, poc_type=0), dict(num_ref=1,
    frames=40, gop=20)])
def test_recovered_chain_launches_are_never_silently_wrong(oracle, kw):
    """when a chain launch is recovered, the lane's NEXT batch has already run and may have decoded into surfaces the redo needs (18
    surfaces round robin, depth-8 chains, several references) or that an IDR / flush displays.  The engine now checks that per decoder: either the redo
    reads intact data -- then every frame equals the oracle's and no error is reported -- or the handle REPORTS errors for what it could not redo.  What
    may not happen is a wrong frame with errors == 0."""
    data = streams.generate(width=352, height=288, seed=0x4D91, mode=1, **kw)
    want, n, w, h = oracle.decode(data, 1)
    fs = w * h * 3 // 2
    lib = api.lib()
    ours = _wait_until_the_gpu_is_ours()
    with api.JmAmdDec(0, 1) as d:
        lib.jm_amddec_set_option(d.h, b"chain_depth", 0)
        lib.jm_amddec_set_option(d.h, b"debug_stall", 2)
        before = d.stat("eng_chain_recoveries")
        try:
            frames = d.decode_stream(None, chunks=[data])               # one chunk: the parser runs far ahead, chains are as deep as they get
            errs, rec = d.stat("errors"), d.stat("eng_chain_recoveries") - before
        finally:
            lib.jm_amddec_set_option(d.h, b"debug_stall", 0)
            lib.jm_amddec_set_option(d.h, b"chain_depth", 0)
    assert len(frames) == n
    wrong = [i for i in range(n) if frames[i] != want[i * fs:(i + 1) * fs]]
    assert not wrong or errs > 0, f"frames {wrong[:8]} differ from the oracle after a recovery and the handle reports no error"
    assert rec >= 1 or not ours, "debug_stall 2 did not make a chain launch give up on a GPU the engine owns"
    _chains_must_have_formed(ours, "the recovery check")


# ---- the "direct" output route (host_copy.cpp): whatever the caller does with its buffers, the bytes are the oracle's -----------------------
def _pull_all(data, w, h, next_buffer, after_frame=None):
    """test_nv_dec's loop with the caller's buffer chosen per frame by next_buffer(i) -> (address, keepalive)."""
    fs = w * h * 3 // 2
    frames = []
    with api.JmAmdDec(0, 1) as d:
        def pull():
            addr, keep = next_buffer(len(frames))
            n = C.c_int(fs)
            ret = api.lib().jm_amddec_output_frame(C.c_void_p(addr), C.byref(n), d.h)
            assert ret == n.value == fs
            frames.append(C.string_at(addr, fs))
            if after_frame:
                after_frame(len(frames) - 1, addr, keep)
        for nal in api.split_nalus(data):
            _, got = api.jm_nvdec_decode_frame(nal, len(nal), d.h)
            if got:
                pull()
        while not api.jm_nvdec_is_exit(d.h):
            _, got = api.jm_nvdec_decode_frame(None, 0, d.h)
            if got:
                pull()
        stats = (d.stat("errors"), d.stat("direct_frames"))
    return frames, stats


_DIRECT_KW = dict(width=320, height=240, frames=12, gop=6, qp=26, seed=0x4D52)


def test_direct_route_into_a_reused_buffer(oracle):
    data = streams.generate(**_DIRECT_KW)
    want = oracle.decode(data, 1)[0]
    buf = C.create_string_buffer(320 * 240 * 3 // 2)
    frames, (errors, direct) = _pull_all(data, 320, 240, lambda i: (C.addressof(buf), buf))
    assert errors == 0 and b"".join(frames) == want
    assert direct == 12                    # every frame went by one copy-engine transfer into the caller's buffer


def test_direct_route_with_a_fresh_buffer_per_frame(oracle):
    data = streams.generate(**_DIRECT_KW)
    want = oracle.decode(data, 1)[0]
    keep = []

    def fresh(i):
        b = C.create_string_buffer(320 * 240 * 3 // 2 + 4096 * (i % 3))
        keep.append(b)
        return C.addressof(b), b
    frames, (errors, direct) = _pull_all(data, 320, 240, fresh)
    assert errors == 0 and b"".join(frames) == want and direct == 12


def test_direct_route_survives_the_caller_unmapping_a_locked_buffer(oracle):
    """The caller unmaps its output buffer and carries on with a new mapping (which the kernel usually places at the same address) every five
    frames.  The buffer is page-locked per call, never across calls, so nothing can go stale: every frame must still be the oracle's.
    (A lock kept between calls made the runtime abort in exactly this test.)"""
    import mmap
    data = streams.generate(**dict(_DIRECT_KW, frames=18))
    want = oracle.decode(data, 1)[0]
    size = (320 * 240 * 3 // 2 + 4095) // 4096 * 4096
    maps = {}

    def buf_for(i):
        k = i // 5                             # a new mapping every five frames; the old one is closed (unmapped) first
        if k not in maps:
            for m in maps.values():
                m[1].close()
            maps.clear()
            m = mmap.mmap(-1, size)
            maps[k] = (C.addressof(C.c_char.from_buffer(m)), m)
        return maps[k][0], maps[k][1]
    frames, (errors, direct) = _pull_all(data, 320, 240, buf_for)
    assert errors == 0 and b"".join(frames) == want and direct == 18


def test_direct_route_two_handles_one_buffer(oracle):
    """A single-threaded application that drives two handles and gives both the same output buffer."""
    a = streams.generate(**_DIRECT_KW)
    b = streams.generate(**dict(_DIRECT_KW, seed=0x4D53))
    wa, wb = oracle.decode(a, 1)[0], oracle.decode(b, 1)[0]
    fs = 320 * 240 * 3 // 2
    buf = C.create_string_buffer(fs)
    got = {0: [], 1: []}
    with api.JmAmdDec(0, 1) as d0, api.JmAmdDec(0, 1) as d1:
        hs = (d0, d1)
        nals = (api.split_nalus(a), api.split_nalus(b))

        def step(k, nal):
            _, g = api.jm_nvdec_decode_frame(nal, len(nal) if nal else 0, hs[k].h)
            if g:
                ret, n = api.jm_nvdec_output_frame(buf, fs, hs[k].h)
                assert ret == n == fs
                got[k].append(buf.raw)
        for i in range(max(len(nals[0]), len(nals[1]))):
            for k in (0, 1):
                if i < len(nals[k]):
                    step(k, nals[k][i])
        for k in (0, 1):
            while not api.jm_nvdec_is_exit(hs[k].h):
                step(k, None)
        assert d0.stat("errors") == 0 and d1.stat("errors") == 0
    assert b"".join(got[0]) == wa and b"".join(got[1]) == wb


def test_direct_route_into_pinned_host_memory(oracle):
    """An application that already page-locked its output buffer with the HIP runtime (hipHostMalloc)."""
    hip = C.CDLL("libamdhip64.so")
    data = streams.generate(**_DIRECT_KW)
    want = oracle.decode(data, 1)[0]
    ptr = C.c_void_p(0)
    assert hip.hipHostMalloc(C.byref(ptr), C.c_size_t(320 * 240 * 3 // 2), C.c_uint(0)) == 0
    try:
        frames, (errors, _) = _pull_all(data, 320, 240, lambda i: (ptr.value, None))
    finally:
        hip.hipHostFree(ptr)
    assert errors == 0 and b"".join(frames) == want


@pytest.mark.parametrize("delay", [1, 3, 8])
def test_display_delay_changes_when_frames_come_out_not_what_they_are(oracle, delay):
    """set_option("display_delay", n) (the reference's ulMaxDisplayDelay, nv_dec.cpp:341): frames are withheld while n pictures are on their way; the end of
    the stream drains everything -- same frames, same order."""
    data = streams.generate(width=176, height=144, frames=14, gop=7, qp=28, num_ref=2, seed=0x4D60, bframes=2, poc_type=0, cabac=1)
    want = oracle.decode(data, 1)[0]
    frames = gpu_decode(data, display_delay=delay)
    assert len(frames) == 14 and b"".join(frames) == want


def test_all_intra_stream_on_device(oracle):
    """an intra-only stream (every picture an I picture) fed in one chunk -- more I pictures in flight than the three worst-case job
    buffers a handle lends: they take ordinary slots, which grow once to I-picture size (decoder.cpp acquire_job_slot).  Bit-exact, no error, a bounded
    number of grow events."""
    data = streams.generate(width=352, height=288, frames=48, gop=1, qp=16, seed=0x4D96)
    want, n, w, h = oracle.decode(data, 1)
    with api.JmAmdDec(0, 1) as d:
        frames = d.decode_stream(None, chunks=[data])
        assert d.stat("errors") == 0 and d.stat("i_pictures") == 48 and d.stat("job_regrown") <= 48
    assert len(frames) == n == 48 and b"".join(frames) == want


def test_many_gpus_in_one_process_mode_on_the_device(oracle, monkeypatch):
    """the drop-in mode 'one process, handles round robin over every GPU' on real hardware.  The box has one GPU, so two handles name
    devices 0 and 1 (device 1 wraps to the GPU) and JM_AMD_DEC_FAKE_NUMA puts them on two NUMA nodes: two parse pools (one per node), page-locked job buffers
    allocated under each node's policy, both handles decoding concurrently through ONE engine -- frames bit-exact, each handle on its node."""
    monkeypatch.setenv("JM_AMD_DEC_FAKE_NUMA", "0:0,1:1")
    monkeypatch.delenv("JM_AMD_DEC_DEVICE", raising=False)
    data = [streams.generate(width=352, height=288, frames=24, gop=12, mode=1, num_ref=2, seed=0x4D97 + k, cabac=k) for k in (0, 1)]
    want = [oracle.decode(d, 1)[0] for d in data]
    got, info = {}, {}

    def run(dev):
        with api.JmAmdDec(0, 1, options={"device": dev}) as d:
            got[dev] = b"".join(d.decode_stream(data[dev]))
            info[dev] = (d.stat("numa_node"), d.stat("device"), d.stat("threads"), d.stat("errors"))
    ts = [threading.Thread(target=run, args=(dev,)) for dev in (0, 1)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for dev in (0, 1):
        assert got[dev] == want[dev], f"handle on device {dev}"
        assert info[dev][0] == dev and info[dev][1] == 0 and info[dev][2] >= 1 and info[dev][3] == 0, info


def test_diagnostic_chain_launches_are_bit_exact_and_print_a_time_line(oracle, tmp_path):
    """Round 4: JM_AMD_DEC_CENSUS turns every chain launch into a diagnostic one (workgroup census, time stamps per picture, wait ticks) and
    JM_AMD_DEC_CHAIN_TIMELINE prints them when the launch retires (Engine::dump_chain_state).  Both are read once per process, hence the child process:
    one stream of 24 frames must decode bit-exactly with them on, and the child reports how many chain launches it saw and their time lines."""
    import subprocess, sys
    data = streams.generate(width=352, height=288, frames=24, gop=12, mode=1, num_ref=1, seed=0x71AE)
    want = oracle.decode(data, 1)[0]
    src, dst = tmp_path / "in.h264", tmp_path / "out.yuv"
    src.write_bytes(data)
    child = (
        "import sys, os\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "from jmcodec_amd import api\n"
        f"data = open({str(src)!r}, 'rb').read()\n"
        "with api.JmAmdDec(0, 1) as d:\n"
        "    out = b''.join(d.decode_stream(data))\n"
        "    print('chains', d.stat('eng_chain_batches'), 'shared', d.stat('eng_gpu_shared'), 'errors', d.stat('errors'))\n"
        f"open({str(dst)!r}, 'wb').write(out)\n")
    # (the parent holds compute queues on the GPU too, so the child's engine would see it as shared and form no chain launches; the parent is idle while
    # it waits for the child, so the child may ignore it)
    env = dict(os.environ, JM_AMD_DEC_CENSUS="1", JM_AMD_DEC_CHAIN_TIMELINE="1", JM_AMD_DEC_IGNORE_SHARED_GPU="1")
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert dst.read_bytes() == want
    words = r.stdout.split()
    chains, shared = int(words[words.index("chains") + 1]), int(words[words.index("shared") + 1])
    assert int(words[words.index("errors") + 1]) == 0
    assert chains >= 1 and shared == 0, r.stdout
    assert r.stderr.count("chain launch of") == chains, r.stderr[-2000:]
    assert "time line picture" in r.stderr and "reconstruction workgroups:" in r.stderr, r.stderr[-2000:]


_QUAD_SCRIPT = r"""
import json, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from jmcodec_amd import api
from tools import streams
out = {}
orc = streams.Oracle()
# search: the generator's integer search range = how far apart the vectors of neighbouring macroblocks can lie.  4: a workgroup's four macroblocks nearly
# always share a window; 24 / 48: many groups sit at the bounds of pk::quad_geometry (vectors ~29 samples / 11 rows apart) or beyond; small pictures: every
# window touches a picture border now and then (the one-window path, and with it the shared window, is then refused per macroblock)
# (mode 1 = random decisions: a fifth of its vectors up to 80 samples long, most macroblocks in several partitions -- those never take the one-window path)
for name, kw in (("near", dict(width=640, height=368, frames=24, gop=24, seed=311, search=4)),
                 ("bounds", dict(width=640, height=368, frames=24, gop=24, seed=312, search=24)),
                 ("far", dict(width=352, height=288, frames=24, gop=24, seed=313, search=48)),
                 ("fuzz", dict(width=352, height=288, frames=24, gop=24, seed=315, mode=1, no_intra=1, num_ref=2)),
                 ("borders", dict(width=96, height=80, frames=24, gop=24, seed=314, search=16))):
    data = streams.generate(**kw)
    want, n, w, h = orc.decode(data, 1)
    with api.JmAmdDec(0, 1) as d:
        q0, p0, c0 = d.stat("eng_quad_windows"), d.stat("eng_private_windows"), d.stat("eng_chain_pics")
        frames = d.decode_stream(None, chunks=[data])
        out[name] = dict(equal=b"".join(frames) == want, frames=len(frames), n=n, errors=d.stat("errors"), quad=d.stat("eng_quad_windows") - q0,
                         private=d.stat("eng_private_windows") - p0, chained=d.stat("eng_chain_pics") - c0, shared=d.stat("eng_gpu_shared"))
print("QUAD " + json.dumps(out))
"""


def test_shared_reference_window_path_ran_and_is_bit_exact(tmp_path):
    """the "quad" path of the chain kernels -- ONE reference window for the four macroblocks of a reconstruction workgroup, a
    data-dependent vote through LDS (recon_device.h) -- must be PROVEN to run on the GPU, not assumed: a diagnostic run (JM_AMD_DEC_CENSUS, its own
    process: the switch is read once) counts the workgroups that shared a window and those that did not, on streams whose vectors lie close together, at
    the bounds of pk::quad_geometry, far apart and against picture borders.  Every stream must equal the oracle; the first must take the shared path,
    the wide-range ones both paths."""
    import json
    import subprocess
    import sys
    from util import ROOT
    ours = _wait_until_the_gpu_is_ours()
    src = tmp_path / "quad.py"
    src.write_text(_QUAD_SCRIPT)
    r = subprocess.run([sys.executable, str(src), ROOT], capture_output=True, text=True, timeout=900, env=dict(os.environ, JM_AMD_DEC_CENSUS="1",
        JM_AMD_DEC_IGNORE_SHARED_GPU="1"))     # (this very process holds queues on the GPU, idle meanwhile: the child must not take it for company)
    line = next((l for l in r.stdout.splitlines() if l.startswith("QUAD ")), None)
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(line[5:])
    for name, v in res.items():
        assert v["equal"] and v["frames"] == v["n"] and v["errors"] == 0, (name, v)
    if not ours or not any(v["chained"] for v in res.values()):
        pytest.skip("no chain launch formed (GPU shared with a third process): the census has nothing to count")
    import util
    util.SESSION_NOTES.append("shared reference windows (quad path) / private windows per stream: " +
                              ", ".join(f"{k} {v['quad']} / {v['private']}" for k, v in res.items()))
    assert res["near"]["quad"] > 0 and res["bounds"]["quad"] > 0, (res["near"], res["bounds"])          # the shared window really was fetched
    assert res["fuzz"]["private"] > 0 and res["borders"]["private"] > 0, (res["fuzz"], res["borders"])  # ... and refused where it must be


# ---- closing test of this file (pytest runs a file's tests in definition order): were the chain kernels exercised at all? ---------------------------
def test_zz_chain_kernels_ran_in_this_session():
    """1-16-stream callers (config C4's 8 streams per GPU, the reference harness's single stream) run on k_chain and
    k_chain_i, so a green suite must prove that both kernels were launched in THIS process -- the engine counts the launches of each (process-wide
    statistics).  Fails when a kernel never ran although the engine owned the GPU at some point; skips (visibly) only when the GPU was shared with
    another process every time a chain test looked.  The counts are printed in the terminal summary (conftest.py)."""
    import util
    with api.JmAmdDec(0, 1) as d:
        d.decode_stream(streams.generate(width=64, height=48, frames=2, gop=2))
        total, with_intra, pics = d.stat("eng_chain_batches"), d.stat("eng_chain_i_batches"), d.stat("eng_chain_pics")
        rec, shared = d.stat("eng_chain_recoveries"), d.stat("eng_gpu_shared")
    util.SESSION_NOTES.append(f"chain launches in this session: k_chain {total - with_intra}, k_chain_i {with_intra} ({pics} pictures; {rec} recovered on "
                              f"purpose by the debug_stall tests); GPU seen unshared in {sum(_GPU_OURS_SEEN)} of {len(_GPU_OURS_SEEN)} looks, "
                              f"shared now: {shared}")
    if not _GPU_OURS_SEEN:
        pytest.skip("no chain test ran in this session (test selection)")
    if not any(_GPU_OURS_SEEN):
        pytest.skip("the GPU was shared with another process at every look: no chain launch could form in this session")
    assert total - with_intra >= 1, "k_chain (P / B chains) never ran in this session"
    assert with_intra >= 1, "k_chain_i (chains with an intra picture) never ran in this session"
