"""BASELINE config 0 at its shape: the push / pull API of /root/reference/intel_dec/jm_intel_dec.h:29-122, driven exactly like the reference's own
harness drives it (test_intel_dec/test_intel_dec.cpp:64-102: while not is_exit: if need_more_data, input_data(up to free_buf_len bytes) -- set_eof when the
file ran out; one output_frame per turn), on the C1 stream (1920x1080 Baseline).

CPU half (no GPU): the call protocol, the hold-off, the frame rate of jm_intel_get_stream_info -- host stages only ("parse_only").
GPU half (-m gpu): the frames themselves, bit-exact against the oracle, through output_frame and through the YUV callback, with pushes of free_buf_len.
"""
import ctypes as C
import os
import subprocess

import pytest

from jmcodec_amd import api
from tools import streams
from util import ROOT, md5

MB = 1 << 20


def test_free_buf_len_is_the_reference_chunk():
    h = api.jm_intel_dec_create_handle()
    assert api.jm_intel_dec_free_buf_len(h) == MB          # the reference's input buffer starts at 1 MB (intel_dec.h)
    api.jm_intel_dec_deinit(h)


@pytest.mark.parametrize("fps", [0, 24, 30, 60])
def test_frame_rate_from_the_vui_h264(fps):
    """dec_get_stream_info (intel_dec.cpp:975-990) divides FrameRateExtN by FrameRateExtD, which Media SDK takes from the VUI timing information:
    time_scale / (2 * num_units_in_tick) for H.264 (E.2.1).  The generator writes tick 1, time_scale 2 * fps; without VUI the rate is 0."""
    data = streams.generate(width=96, height=80, frames=5, gop=5, vui_fps=fps)
    frames, info, (ret, w, h, rate), _ = api.intel_push_pull(data, options={"parse_only": 1})
    assert (ret, w, h) == (0, 96, 80) and len(frames) == 5
    assert rate == float(fps)


def test_vui_does_not_change_the_decoded_frames(oracle):
    """The VUI sits behind everything that influences decoding: same frames with and without it (oracle, which skips it, and generator agree)."""
    kw = dict(width=96, height=80, frames=4, gop=4, mode=1, seed=91)
    a = oracle.decode(streams.generate(**kw), 1)
    b = oracle.decode(streams.generate(vui_fps=25, **kw), 1)
    assert a == b


def test_frame_rate_from_the_vui_hevc():
    for fps in (0, 50):
        data = streams.generate_hevc(width=128, height=96, frames=4, vui_fps=fps)
        frames, info, (ret, w, h, rate), _ = api.intel_push_pull(data, codec_type=1, options={"parse_only": 1})
        assert (ret, w, h, len(frames)) == (0, 128, 96, 4)
        assert rate == float(fps)
    want = streams.OracleHevc().decode(streams.generate_hevc(width=128, height=96, frames=4), 1)
    assert streams.OracleHevc().decode(streams.generate_hevc(width=128, height=96, frames=4, vui_fps=50), 1) == want


def test_third_party_clip_through_the_push_pull_loop():
    """imageio's sample clip (tests/golden/thirdparty.json): a VUI written by a foreign encoder -- it carries a bitstream restriction and no timing
    information (SPS 27 64 00 28 ac 2b 40 a0 fd 00 f1 22 6a: timing_info_present_flag 0), so the rate is reported as 0."""
    data = open(os.path.join(ROOT, "tests", "golden", "thirdparty_realshort.h264"), "rb").read()
    frames, info, (ret, w, h, rate), _ = api.intel_push_pull(data, options={"parse_only": 1})
    assert (ret, w, h, len(frames), rate) == (0, 320, 240, 36, 0.0)


def test_push_pull_loop_host_side_with_whole_buffer_pushes():
    """The reference loop with pushes of free_buf_len: a 1 MB push holds MANY small pictures, far more than one output_frame call per turn takes out.
    Every frame must come out, in order, and the loop must end (parse-only frames carry no samples: count + protocol only)."""
    data = streams.generate(width=176, height=144, frames=120, gop=30, seed=12)
    assert len(data) < MB                                    # the whole stream goes in with the first push
    frames, info, sinfo, biggest = api.intel_push_pull(data, options={"parse_only": 1})
    assert len(frames) == 120 and biggest == len(data)
    assert "Frame Count:\t120" in info and "Display:\t176 x 144" in info


def test_input_buffer_and_feeder_hold_off():
    """The facade's input side (jm_intel_dec_api.cpp): input_data copies into a 1 MB buffer that a feeder thread hands to the decoder; need_more_data /
    free_buf_len describe that buffer (intel_dec.cpp:343-360); the feeder stops while 64 display frames wait for the caller, so a caller that does not
    fetch cannot make the decoder run through the whole stream; size query and a too-small buffer leave the frame where it is."""
    import time
    data = streams.generate(width=96, height=80, frames=300, gop=30, seed=3)
    L = api.lib()
    h = api.jm_intel_dec_create_handle()
    dec = L.jm_amdintel_decoder(h)
    L.jm_amddec_set_option(dec, b"parse_only", 1)
    assert api.jm_intel_dec_init(0, 1, h) == 0
    assert api.jm_intel_dec_need_more_data(h) and api.jm_intel_dec_free_buf_len(h) == MB
    assert 64 * 1024 < len(data) < MB // 2                                       # more than one of the feeder's 64 KB pieces
    assert api.jm_intel_dec_input_data(data, len(data), h) == len(data)        # one push of the whole stream: accepted at once
    for _ in range(2000):                                                        # the feeder works through it until 64 frames wait
        if L.jm_amddec_get_stat(dec, b"frames_waiting") >= 64:
            break
        time.sleep(0.001)
    time.sleep(0.05)
    waiting, pics = L.jm_amddec_get_stat(dec, b"frames_waiting"), L.jm_amddec_get_stat(dec, b"pictures")
    assert 64 <= waiting < 300 and pics < 300, (waiting, pics)                   # held back: a 64 KB piece beyond the limit at most
    # (the engine's urgency rule reads this count: display frames whose samples are there and that nobody has fetched -- in parse-only mode every queued frame)
    assert L.jm_amddec_get_stat(dec, b"frames_done_unfetched") == waiting
    assert api.jm_intel_dec_free_buf_len(h) < MB                                 # ... with input still in the buffer
    out = C.create_string_buffer(96 * 80 * 3 // 2)
    # a size query and a too-small buffer leave the frame where it is (jm_intel_dec.h:69-78, intel_dec.cpp:266-270)
    assert api.jm_intel_dec_output_frame(None, 0, h) == (0, len(out))
    assert api.jm_intel_dec_output_frame(out, 10, h) == (-2, 0)
    assert api.jm_intel_dec_output_frame(out, len(out), h) == (0, len(out))
    got = 1
    api.jm_intel_dec_set_eof(1, h)
    assert not api.jm_intel_dec_need_more_data(h)
    assert api.jm_intel_dec_input_data(data[:100], 100, h) < 0                   # input after end of stream is refused
    guard = 0
    while not api.jm_intel_dec_is_exit(h):
        guard += 1
        assert guard < 10_000_000
        if api.jm_intel_dec_output_frame(out, len(out), h)[0] == 0:
            got += 1
    assert got == 300
    assert L.jm_amddec_get_stat(dec, b"frames_done_unfetched") == 0
    assert "Frame Count:\t300" in api.jm_intel_dec_info(h)
    api.jm_intel_dec_deinit(h)


def test_deinit_with_input_still_buffered():
    """A handle closed in the middle of everything: input buffered, the feeder waiting for the caller, frames never fetched."""
    data = streams.generate(width=96, height=80, frames=200, gop=40, seed=4)
    for fetch in (0, 3):
        h = api.jm_intel_dec_create_handle()
        api.lib().jm_amddec_set_option(api.lib().jm_amdintel_decoder(h), b"parse_only", 1)
        assert api.jm_intel_dec_init(0, 1, h) == 0
        assert api.jm_intel_dec_input_data(data, len(data), h) == len(data)
        out = C.create_string_buffer(96 * 80 * 3 // 2)
        for _ in range(fetch):
            while api.jm_intel_dec_output_frame(out, len(out), h)[0] != 0:
                pass
        assert api.jm_intel_dec_deinit(h) == 0


def test_native_push_pull_loop_host_side():
    data = streams.generate(width=176, height=144, frames=50, gop=25, seed=5)
    L = api.lib()
    h = api.jm_intel_dec_create_handle()
    L.jm_amddec_set_option(L.jm_amdintel_decoder(h), b"parse_only", 1)
    assert api.jm_intel_dec_init(0, 1, h) == 0
    out = (C.c_ubyte * (176 * 144 * 3 // 2))()
    assert L.jm_amdintel_run_pushpull(data, len(data), out, len(out), h) == 50
    assert api.jm_intel_dec_is_exit(h)
    api.jm_intel_dec_deinit(h)


# ---------------------------------------------------------------------------------------------------------------------------------------
# GPU half
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c0_stream():
    """SURVEY 8d C0 = the C1 stream (stream_id 0); 60 frames = two IDR periods = 1.3 MB (22 KB per frame at QP 28: a push of free_buf_len = 1 MB holds
    about 45 pictures), with VUI timing so that the frame rate has an answer."""
    return streams.generate(vui_fps=30, **streams.config_c1(frames=60))


@pytest.mark.gpu
@pytest.mark.parametrize("callback", [False, True])
def test_c0_push_pull_1080p_vs_oracle(oracle, c0_stream, callback):
    """60 frames of the C1 stream through the reference's loop with pushes of min(free_buf_len, remaining) = 1 MB (the first push alone holds some 45
    pictures): bit-exact against the oracle, once through output_frame and once through the YUV callback."""
    want, n, w, h = oracle.decode(c0_stream, 1)
    assert (n, w, h) == (60, 1920, 1080)
    fs = w * h * 3 // 2
    assert MB < len(c0_stream) < 2 * MB                    # two pushes: one of free_buf_len, one of the rest
    frames, info, (ret, sw, sh, rate), biggest = api.intel_push_pull(c0_stream, callback=callback)
    assert biggest == MB
    assert (ret, sw, sh, rate) == (0, 1920, 1080, 30.0)
    assert len(frames) == n
    for i, f in enumerate(frames):
        assert f == want[i * fs:(i + 1) * fs], f"frame {i} differs from the oracle"
    assert "Frame Count:\t60" in info and "Display:\t1920 x 1080" in info


@pytest.mark.gpu
def test_c0_push_pull_300_frames_properties(c0_stream):
    """C0's 300 frames (the oracle would take a quarter of a minute per pass): five repetitions of the same two closed GOPs decode to the same sixty
    frames each, and to the frames the NAL-per-call API (jm_nvdec_*) gives for the same stream; the native loop (jm_amdintel_run_pushpull, what bench.py
    times) returns the same count."""
    data = c0_stream * 5
    digests = []
    frames, info, sinfo, biggest = api.intel_push_pull(data, on_frame=lambda f: digests.append(md5(f)))
    assert len(digests) == 300 and biggest == MB
    for g in range(1, 5):
        assert digests[60 * g:60 * g + 60] == digests[:60]
    assert len(set(digests[:60])) == 60
    with api.JmAmdDec(0, 1) as d:
        ref = [md5(f) for f in d.decode_stream(c0_stream)]
    assert ref == digests[:60]
    L = api.lib()
    h = api.jm_intel_dec_create_handle()
    assert api.jm_intel_dec_init(0, 1, h) == 0
    out = (C.c_ubyte * (1920 * 1080 * 3 // 2))()
    assert L.jm_amdintel_run_pushpull(data, len(data), out, len(out), h) == 300
    assert md5(bytes(out)) == digests[59]                   # the last frame fetched sits in the caller's buffer
    assert L.jm_amddec_get_stat(L.jm_amdintel_decoder(h), b"errors") == 0
    # one copy per frame: every frame left by the copy-engine route straight into the caller's buffer
    assert L.jm_amddec_get_stat(L.jm_amdintel_decoder(h), b"direct_frames") == 300
    api.jm_intel_dec_deinit(h)


@pytest.mark.gpu
def test_c0_push_pull_hevc_1080p_vs_oracle():
    """The same loop with codec_type 1 at 1080p (the facade had only ever seen 128x96 HEVC)."""
    data = streams.generate_hevc(vui_fps=60, **streams.config_c3(frames=9, width=1920, height=1080))
    want, n, w, h = streams.OracleHevc().decode(data, 1)
    fs = w * h * 3 // 2
    frames, info, (ret, sw, sh, rate), biggest = api.intel_push_pull(data, codec_type=1)
    assert (ret, sw, sh, rate) == (0, 1920, 1080, 60.0)
    assert len(frames) == n
    for i, f in enumerate(frames):
        assert f == want[i * fs:(i + 1) * fs], f"frame {i} differs from the oracle"


def _ref_harness():
    return os.path.join(ROOT, "oracle", "_ref", "test_intel_dec")


@pytest.mark.gpu
def test_reference_test_intel_dec_binary_on_the_device(c0_stream, tmp_path):
    """The reference's own harness (test_intel_dec/test_intel_dec.cpp, compiled IN PLACE by oracle/Makefile into oracle/_ref/ -- never copied into the
    repo, SURVEY 8b link proof) run against the library on the GPU with the C0 stream: its loop must end and report all 300 frames at 1920 x 1080."""
    exe = _ref_harness()
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/test_intel_dec was not built (the reference tree exists only in the build container)")
    src = tmp_path / "c0.h264"
    src.write_bytes(c0_stream * 5)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(api.lib_path()) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(src)], capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Frame Count:\t300" in r.stdout and "Display:\t1920 x 1080" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
