"""ctypes binding of ``include/jm_amd_dec.h`` plus a Python mirror of the reference API.

Mirrors /root/reference/nv_dec/jm_nv_dec.h:27-88 (``jm_nvdec_*``) and the call loop of
/root/reference/test_nv_dec/test_nv_dec.cpp:163-259.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIB = None


def lib_path():
    # JM_AMD_DEC_LIB: developer aid (timing experiments with alternative builds of the library); the product is lib/libjm_amd_dec.so
    return os.environ.get("JM_AMD_DEC_LIB") or os.path.join(_HERE, "lib", "libjm_amd_dec.so")


def build(force=False):
    """Compile the HIP/C++ sources in-tree (hipcc --offload-arch=gfx950)."""
    if force or not os.path.exists(lib_path()) or _stale():
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], stdout=subprocess.DEVNULL)
    return lib_path()


def _stale():
    try:
        t = os.path.getmtime(lib_path())
        src = os.path.join(_HERE, "csrc")
        return any(os.path.getmtime(os.path.join(src, f)) > t for f in os.listdir(src))
    except OSError:
        return True


def lib():
    """Load libjm_amd_dec.so; raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run jmcodec_amd.build() / __graft_entry__.build() first "
                           "(the HIP backend is mandatory, there is no CPU fallback)")
    L = C.CDLL(path)
    vp, cp, ip = C.c_void_p, C.c_char_p, C.POINTER(C.c_int)
    L.jm_amddec_create_handle.restype = vp
    L.jm_amddec_init.argtypes = [C.c_int, C.c_int, cp, C.c_int, vp]
    L.jm_amddec_deinit.argtypes = [vp]
    L.jm_amddec_decode_frame.argtypes = [vp, C.c_int, ip, vp]
    L.jm_amddec_output_frame.argtypes = [vp, ip, vp]
    L.jm_amddec_stream_info.argtypes = [ip, ip, vp]
    L.jm_amddec_set_eof.argtypes = [C.c_int, vp]
    L.jm_amddec_set_eof.restype = None
    L.jm_amddec_is_exit.argtypes = [vp]
    L.jm_amddec_show_dec_info.argtypes = [vp]
    L.jm_amddec_show_dec_info.restype = cp
    L.jm_amddec_set_option.argtypes = [vp, cp, C.c_longlong]
    L.jm_amddec_get_stat.argtypes = [vp, cp]
    L.jm_amddec_get_stat.restype = C.c_longlong
    L.jm_amddec_last_error.argtypes = [vp]
    L.jm_amddec_last_error.restype = cp
    L.jm_amddec_packout_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.jm_amddec_output_frame_device.argtypes = [C.POINTER(C.c_void_p), ip, vp]
    L.jm_amddec_feed_annexb.argtypes = [cp, C.c_long, C.c_int, C.POINTER(C.c_ubyte), C.c_int, vp]
    L.jm_amddec_feed_annexb.restype = C.c_long
    L.jm_amddec_poll_frame.argtypes = [ip, vp]
    L.jm_amddec_wait_frame.argtypes = [ip, C.c_int, vp]
    L.jm_amddec_push_data.argtypes = [vp, C.c_int, vp]
    # push / pull API (include/jm_amd_intel_dec.h <- /root/reference/intel_dec/jm_intel_dec.h:29-122)
    L.jm_amdintel_create_handle.restype = vp
    L.jm_amdintel_init.argtypes = [C.c_int, C.c_int, vp]
    L.jm_amdintel_deinit.argtypes = [vp]
    L.jm_amdintel_set_yuv_callback.argtypes = [vp, vp, vp]
    L.jm_amdintel_input_data.argtypes = [vp, C.c_int, vp]
    L.jm_amdintel_output_frame.argtypes = [vp, ip, vp]
    L.jm_amdintel_set_eof.argtypes = [C.c_int, vp]
    L.jm_amdintel_info.argtypes = [vp]
    L.jm_amdintel_info.restype = cp
    L.jm_amdintel_get_stream_info.argtypes = [ip, ip, C.POINTER(C.c_float), vp]
    L.jm_amdintel_need_more_data.argtypes = [vp]
    L.jm_amdintel_free_buf_len.argtypes = [vp]
    L.jm_amdintel_is_exit.argtypes = [vp]
    L.jm_amdintel_run_pushpull.argtypes = [cp, C.c_long, C.POINTER(C.c_ubyte), C.c_int, vp]
    L.jm_amdintel_run_pushpull.restype = C.c_long
    L.jm_amdintel_decoder.argtypes = [vp]
    L.jm_amdintel_decoder.restype = vp
    _LIB = L
    return L


# ---- the reference's ten functions, same names and argument order -------------------------
def jm_nvdec_is_hw_support():
    return bool(lib().jm_amddec_is_hw_support())


def jm_nvdec_create_handle():
    return lib().jm_amddec_create_handle()


def jm_nvdec_init(codec_type, out_fmt, extra_data, length, handle):
    return lib().jm_amddec_init(codec_type, out_fmt, extra_data, length, handle)


def jm_nvdec_deinit(handle):
    return lib().jm_amddec_deinit(handle)


def jm_nvdec_decode_frame(in_buf, in_data_len, handle):
    """Returns (ret, got_frame).  in_buf: bytes / ctypes buffer / None."""
    got = C.c_int(0)
    if in_buf is None or in_data_len == 0:
        ret = lib().jm_amddec_decode_frame(None, 0, C.byref(got), handle)
    else:
        if isinstance(in_buf, (bytes, bytearray)):
            in_buf = C.cast(C.c_char_p(bytes(in_buf)), C.c_void_p)
        ret = lib().jm_amddec_decode_frame(in_buf, in_data_len, C.byref(got), handle)
    return ret, got.value


def jm_nvdec_output_frame(out_buf, out_len, handle):
    """out_buf: writable ctypes buffer; out_len: its capacity.  Returns (ret, bytes_written)."""
    n = C.c_int(out_len)
    ret = lib().jm_amddec_output_frame(C.cast(out_buf, C.c_void_p), C.byref(n), handle)
    return ret, n.value


def jm_nvdec_stream_info(handle):
    w, h = C.c_int(0), C.c_int(0)
    lib().jm_amddec_stream_info(C.byref(w), C.byref(h), handle)
    return w.value, h.value


def jm_nvdec_set_eof(is_eof, handle):
    lib().jm_amddec_set_eof(1 if is_eof else 0, handle)


def jm_nvdec_is_exit(handle):
    return bool(lib().jm_amddec_is_exit(handle))


def jm_nvdec_show_dec_info(handle):
    return lib().jm_amddec_show_dec_info(handle).decode()


# ---- the reference harness's NAL scanner (test_nv_dec.cpp:30-86), vectorised -----------------
def split_nalus(data):
    """Split an Annex-B buffer into chunks exactly as test_nv_dec's find_nalu does: each chunk
    starts at a start code (00 00 01 or 00 00 00 01) and runs up to the next one."""
    import numpy as np
    a = np.frombuffer(data, dtype=np.uint8)
    if len(a) < 4:
        return [bytes(data)]
    m = (a[:-2] == 0) & (a[1:-1] == 0) & (a[2:] == 1)
    idx = np.flatnonzero(m)
    starts = []
    for i in idx:
        s = int(i)
        if s > 0 and a[s - 1] == 0:
            s -= 1                      # 4-byte start code: find_nalu_prefix matches it one byte earlier
        if not starts or s > starts[-1]:
            starts.append(s)
    out = []
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(a)
        out.append(bytes(data[s:e]))
    return out


def annexb_to_avcc(data, length_size=4):
    """(avcC record, [length-prefixed packets]) of an Annex-B stream: what a demuxer hands test_player for an MP4 source when no
    h264_mp4toannexb filter is in the way (test_player.cpp:221-226).  One packet per access unit (split before each first slice)."""
    nalus = [n.lstrip(b"\x00")[1:] for n in split_nalus(data)]              # strip start codes
    sps = [n for n in nalus if n and (n[0] & 31) == 7]
    pps = [n for n in nalus if n and (n[0] & 31) == 8]
    rec = bytes([1, sps[0][1], sps[0][2], sps[0][3], 0xFC | (length_size - 1), 0xE0 | 1]) + len(sps[0]).to_bytes(2, "big") + sps[0]
    rec += bytes([1]) + len(pps[0]).to_bytes(2, "big") + pps[0]
    packets, cur = [], b""
    for n in nalus:
        t = n[0] & 31
        if t in (7, 8):
            continue
        if t in (1, 5) and (n[1] & 0x80) and cur:                            # first_mb_in_slice == 0: a new picture starts
            packets.append(cur)
            cur = b""
        cur += len(n).to_bytes(length_size, "big") + n
    if cur:
        packets.append(cur)
    return rec, packets


def annexb_to_hvcc(data, length_size=4):
    """(hvcC record, [length-prefixed packets]) of an HEVC Annex-B stream: the MP4 / MKV form of the same stream (one packet per access
    unit, parameter sets only in the record)."""
    nalus = [n.lstrip(b"\x00")[1:] for n in split_nalus(data)]
    ps = {32: [], 33: [], 34: []}
    for n in nalus:
        t = (n[0] >> 1) & 63
        if t in ps and n not in ps[t]:
            ps[t].append(n)
    rec = bytes([1]) + bytes(20) + bytes([0xFC | (length_size - 1), sum(1 for t in ps if ps[t])])
    for t in (32, 33, 34):
        if ps[t]:
            rec += bytes([0x80 | t]) + len(ps[t]).to_bytes(2, "big") + b"".join(len(n).to_bytes(2, "big") + n for n in ps[t])
    packets, cur = [], b""
    for n in nalus:
        t = (n[0] >> 1) & 63
        if t in ps:
            continue
        if (t <= 9 or 16 <= t <= 21) and (n[2] & 0x80) and cur:            # first_slice_segment_in_pic_flag: a new picture starts
            packets.append(cur)
            cur = b""
        cur += len(n).to_bytes(length_size, "big") + n
    if cur:
        packets.append(cur)
    return rec, packets


class JmAmdDec:
    """Convenience wrapper reproducing test_nv_dec's main loop (test_nv_dec.cpp:163-259)."""

    def __init__(self, codec_type=0, out_fmt=1, options=None, extra_data=None):
        self.h = jm_nvdec_create_handle()
        for k, v in (options or {}).items():
            lib().jm_amddec_set_option(self.h, k.encode(), int(v))
        rc = jm_nvdec_init(codec_type, out_fmt, extra_data, len(extra_data) if extra_data else 0, self.h)
        if rc != 0:
            err = lib().jm_amddec_last_error(self.h).decode()
            jm_nvdec_deinit(self.h)
            self.h = None
            raise RuntimeError(f"jm_nvdec_init failed: {err}")
        self.out_buf = None
        self.nalu_count = 0

    def stat(self, key):
        return lib().jm_amddec_get_stat(self.h, key.encode())

    def _pull(self, frames):
        w, h = jm_nvdec_stream_info(self.h)                    # size of the frame about to be fetched (it can change at an IDR picture)
        if self.out_buf is None or len(self.out_buf) < w * h * 3 // 2:
            self.out_buf = C.create_string_buffer(max(w * h * 3 // 2, 16))
        ret, n = jm_nvdec_output_frame(self.out_buf, len(self.out_buf), self.h)
        if n > 0 and frames is not None:
            frames.append(self.out_buf.raw[:n])
        return n > 0

    def decode_stream(self, data, keep=True, chunks=None):
        """Feed one NAL per call, then drain with (NULL, 0) until is_exit.  Returns list of frames
        (bytes) when keep, else the frame count."""
        frames = [] if keep else None
        count = 0
        for nal in (chunks if chunks is not None else split_nalus(data)):
            self.nalu_count += 1
            _, got = jm_nvdec_decode_frame(nal, len(nal), self.h)
            if got == 1 and self._pull(frames):
                count += 1
        while not jm_nvdec_is_exit(self.h):
            self.nalu_count += 1
            ret, got = jm_nvdec_decode_frame(None, 0, self.h)
            if ret != 0:
                raise RuntimeError(lib().jm_amddec_last_error(self.h).decode())
            if got == 1 and self._pull(frames):
                count += 1
        return frames if keep else count

    def close(self):
        if self.h:
            jm_nvdec_deinit(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ---- the reference's push / pull API (intel_dec/jm_intel_dec.h:29-122), same names and argument order ----
YUV_CALLBACK = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_void_p)     # HANDLE_YUV_CALLBACK, jm_intel_dec.h:20


def jm_intel_is_hw_support():
    return bool(lib().jm_amdintel_is_hw_support())


def jm_intel_dec_create_handle():
    return lib().jm_amdintel_create_handle()


def jm_intel_dec_init(codec_type, out_fmt, handle):
    return lib().jm_amdintel_init(codec_type, out_fmt, handle)


def jm_intel_dec_deinit(handle):
    return lib().jm_amdintel_deinit(handle)


def jm_intel_dec_set_yuv_callback(user_data, callback, handle):
    """callback: a YUV_CALLBACK instance (the caller keeps it alive) or None."""
    return lib().jm_amdintel_set_yuv_callback(user_data, C.cast(callback, C.c_void_p) if callback else None, handle)


def jm_intel_dec_input_data(in_buf, in_data_len, handle):
    """in_buf: bytes or a ctypes buffer / address.  Returns the bytes accepted (> 0) or < 0."""
    if isinstance(in_buf, (bytes, bytearray)):
        in_buf = C.cast(C.c_char_p(bytes(in_buf)), C.c_void_p)
    return lib().jm_amdintel_input_data(in_buf, in_data_len, handle)


def jm_intel_dec_output_frame(out_buf, out_len, handle):
    """out_buf: writable ctypes buffer or None (size query); out_len: its capacity.  Returns (ret, bytes): ret 0 = a frame was copied,
    -1 = none ready, -2 = buffer too small (jm_intel_dec.h:69-78)."""
    n = C.c_int(out_len)
    ret = lib().jm_amdintel_output_frame(C.cast(out_buf, C.c_void_p) if out_buf is not None else None, C.byref(n), handle)
    return ret, n.value


def jm_intel_dec_set_eof(is_eof, handle):
    return lib().jm_amdintel_set_eof(int(is_eof), handle)


def jm_intel_dec_info(handle):
    s = lib().jm_amdintel_info(handle)
    return s.decode() if s else ""


def jm_intel_get_stream_info(handle):
    """Returns (ret, width, height, frame_rate)."""
    w, h, f = C.c_int(0), C.c_int(0), C.c_float(0.0)
    ret = lib().jm_amdintel_get_stream_info(C.byref(w), C.byref(h), C.byref(f), handle)
    return ret, w.value, h.value, f.value


def jm_intel_dec_need_more_data(handle):
    return bool(lib().jm_amdintel_need_more_data(handle))


def jm_intel_dec_free_buf_len(handle):
    return lib().jm_amdintel_free_buf_len(handle)


def jm_intel_dec_is_exit(handle):
    return bool(lib().jm_amdintel_is_exit(handle))


def intel_push_pull(data, codec_type=0, out_fmt=1, callback=False, max_push=None, on_frame=None, options=None):
    """The loop of /root/reference/test_intel_dec/test_intel_dec.cpp:64-102 over an Annex-B buffer: while not is_exit: if need_more_data and input is
    left, input_data(min(free_buf_len, remaining)) -- set_eof when it ran out; then one output_frame.  Frames go to ``on_frame(bytes)`` (default: a
    list that is returned), through output_frame or -- callback=True -- through the YUV callback.  Returns (frames, info string, stream info tuple,
    largest single push)."""
    frames = []
    sink = on_frame or frames.append
    cb = YUV_CALLBACK(lambda p, n, u: sink(C.string_at(p, n)) or 0)
    h = jm_intel_dec_create_handle()
    for k, v in (options or {}).items():
        lib().jm_amddec_set_option(lib().jm_amdintel_decoder(h), k.encode(), int(v))
    if jm_intel_dec_init(codec_type, out_fmt, h) != 0:
        jm_intel_dec_deinit(h)
        raise RuntimeError("jm_intel_dec_init failed")
    try:
        if callback:
            jm_intel_dec_set_yuv_callback(None, cb, h)
        data = bytes(data)
        mv = memoryview(data)
        base = C.cast(C.c_char_p(data), C.c_void_p).value
        out = C.create_string_buffer(20 << 20)                       # the harness's 20 MB output buffer, test_intel_dec.cpp:50
        pos, is_eof, biggest, guard, sinfo = 0, False, 0, 0, None
        while not jm_intel_dec_is_exit(h):
            guard += 1
            if guard > 50_000_000:
                raise RuntimeError("push / pull loop does not terminate")
            if jm_intel_dec_need_more_data(h) and not is_eof:
                k = min(jm_intel_dec_free_buf_len(h), len(mv) - pos)
                if max_push:
                    k = min(k, max_push)
                if k == 0:
                    is_eof = True
                    jm_intel_dec_set_eof(1, h)
                else:
                    if jm_intel_dec_input_data(base + pos, k, h) != k:
                        raise RuntimeError("jm_intel_dec_input_data failed")
                    pos += k
                    biggest = max(biggest, k)
            ret, n = jm_intel_dec_output_frame(out, len(out), h)
            if ret == 0:
                if callback:
                    raise RuntimeError("output_frame handed out a frame although a callback is set")
                sink(C.string_at(out, n))
                if sinfo is None:
                    sinfo = jm_intel_get_stream_info(h)
        if sinfo is None:
            sinfo = jm_intel_get_stream_info(h)
        return frames, jm_intel_dec_info(h), sinfo, biggest
    finally:
        jm_intel_dec_deinit(h)
