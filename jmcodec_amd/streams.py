"""Seeded synthetic Annex-B streams (tools/h264gen.c) and the CPU-oracle binding used by tests/bench.

Neither is part of the decode product: the generator makes inputs (the reference ships none,
SURVEY.md section 4) and ``Oracle`` wraps oracle/ for the checker / cpu_baseline legs only.
"""
import ctypes as C
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class GenParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "width", "height", "frames", "qp", "gop", "seed", "mode", "deblock", "num_ref", "slices",
        "pcm_only", "poc_type", "nonref_period", "alpha_off", "beta_off", "chroma_qp_off", "level_idc",
        "cip", "search", "cabac", "cabac_idc", "t8x8", "bframes", "direct_temporal", "wp", "dinf8", "scaling", "rplm", "mmco", "nc_corner")]


def build_tools():
    subprocess.check_call(["make", "-C", os.path.join(_ROOT, "tools")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


_gen = None


def _genlib():
    global _gen
    if _gen is None:
        p = os.path.join(_ROOT, "tools", "_build", "libh264gen.so")
        if not os.path.exists(p):
            build_tools()
        _gen = C.CDLL(p)
        _gen.h264gen_generate.argtypes = [C.POINTER(GenParams), C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t), C.c_char_p]
        _gen.h264gen_free.argtypes = [C.c_void_p]
    return _gen


def generate(width=64, height=48, frames=4, qp=28, gop=30, seed=0x4A4D0100, mode=0, deblock=1, num_ref=1,
             slices=1, pcm_only=0, poc_type=2, nonref_period=0, alpha_off=0, beta_off=0, chroma_qp_off=0,
             level_idc=0, cip=0, search=4, cabac=0, cabac_idc=0, t8x8=0, bframes=0, direct_temporal=0, wp=0, dinf8=1, scaling=0, rplm=0, mmco=0, nc_corner=0,
             recon_path=None):
    """Returns the Annex-B stream as bytes (optionally writing the encoder's own reconstruction)."""
    p = GenParams(width, height, frames, qp, gop, seed, mode, deblock, num_ref, slices, pcm_only, poc_type,
                  nonref_period, alpha_off, beta_off, chroma_qp_off, level_idc, cip, search, cabac, cabac_idc, t8x8,
                  bframes, direct_temporal, wp, dinf8, scaling, rplm, mmco, nc_corner)
    buf = C.POINTER(C.c_ubyte)()
    n = C.c_size_t(0)
    rc = _genlib().h264gen_generate(C.byref(p), C.byref(buf), C.byref(n), recon_path.encode() if recon_path else None)
    if rc != 0:
        raise ValueError("h264gen: bad parameters")
    data = C.string_at(buf, n.value)
    _genlib().h264gen_free(buf)
    return data


# BASELINE.json configs restated as generator parameters; seed = 0x4A4D0000 + config*256 + stream
def config_c1(stream_id=0, frames=300, width=1920, height=1080):
    return dict(width=width, height=height, frames=frames, qp=28, gop=30, seed=0x4A4D0000 + 1 * 256 + stream_id,
                mode=0, deblock=1, num_ref=1, level_idc=40)


def config_c2(stream_id=0, frames=120, width=3840, height=2160):
    """BASELINE config 2: H.264 High 4K, CABAC, 8x8 transform, I B B P with two references."""
    return dict(width=width, height=height, frames=frames, qp=30, gop=30, seed=0x4A4D0000 + 2 * 256 + stream_id,
                mode=0, deblock=1, num_ref=2, level_idc=52, cabac=1, t8x8=1, bframes=2, poc_type=0)


class Oracle:
    """ctypes binding of oracle/_build/liborc.so (CPU oracle -- checker / cpu_baseline only)."""

    def __init__(self):
        p = os.path.join(_ROOT, "oracle", "_build", "liborc.so")
        if not os.path.exists(p):
            build_tools()
        L = C.CDLL(p)
        L.orc_decode_stream_to_buffer.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.POINTER(C.c_ubyte)),
                                                  C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_open.restype = C.c_void_p
        L.orc_open.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_close.argtypes = [C.c_void_p]
        L.orc_decode_annexb.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_flush.argtypes = [C.c_void_p]
        L.orc_digest_enable.argtypes = [C.c_void_p]
        L.orc_digest_value.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_digest_value.restype = C.c_uint64
        L.orc_last_error.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_packout.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int)]
        self.L = L

    def decode(self, data, out_fmt=1):
        """Returns (frames_bytes, n_frames, width, height): all display-order frames concatenated."""
        buf = C.POINTER(C.c_ubyte)()
        n = C.c_size_t(0)
        w, h = C.c_int(0), C.c_int(0)
        cnt = self.L.orc_decode_stream_to_buffer(data, len(data), out_fmt, C.byref(buf), C.byref(n), C.byref(w), C.byref(h))
        if cnt < 0:
            raise RuntimeError("oracle decode failed")
        out = C.string_at(buf, n.value) if n.value else b""
        self.L.orc_free(buf)
        return out, cnt, w.value, h.value

    def syntax_digest(self, data):
        """(digest, n_macroblocks) of the parsed syntax elements (see oracle/orc_slice.c digest_mb)."""
        d = self.L.orc_open(None, None)
        self.L.orc_digest_enable(d)
        rc = self.L.orc_decode_annexb(d, data, len(data))
        err = self.L.orc_last_error(d).decode()
        self.L.orc_flush(d)
        n = C.c_uint64(0)
        v = self.L.orc_digest_value(d, C.byref(n))
        self.L.orc_close(d)
        if rc < 0:
            raise RuntimeError("oracle: " + err)
        return v, n.value

    def tools(self, data):
        """Which coding tools the stream exercises: {counter name: macroblocks / slices} (non-zero entries)."""
        L = self.L
        L.orc_tool_name.restype = C.c_char_p
        L.orc_tool_name.argtypes = [C.c_int]
        L.orc_tool_count.restype = C.c_long
        L.orc_tool_count.argtypes = [C.c_void_p, C.c_int]
        d = L.orc_open(None, None)
        rc = L.orc_decode_annexb(d, data, len(data))
        err = L.orc_last_error(d).decode()
        L.orc_flush(d)
        out, i = {}, 0
        while L.orc_tool_name(i):
            if L.orc_tool_count(d, i):
                out[L.orc_tool_name(i).decode()] = L.orc_tool_count(d, i)
            i += 1
        L.orc_close(d)
        if rc < 0:
            raise RuntimeError("oracle: " + err)
        return out

    def packout(self, src, pitch, width, height, out_fmt):
        cap = width * height * 3 // 2
        dst = C.create_string_buffer(cap)
        n = C.c_int(cap)
        rc = self.L.orc_packout(src, pitch, width, height, out_fmt, dst, C.byref(n))
        return rc, dst.raw[:n.value]


# ---------------------------------------------------------------------------------------------------------
# HEVC: tools/hevcgen.c (stream generator) and oracle/orc_hevc_*.c (CPU oracle) -- test infrastructure
# ---------------------------------------------------------------------------------------------------------
HEVC_FIELDS = ("width", "height", "frames", "qp", "seed", "intra_period", "gop", "num_ref", "ctb_log2", "min_cb_log2", "max_tb_log2", "min_tb_log2",
               "depth_inter", "depth_intra", "mode", "amp", "sao", "deblock", "tskip", "sdh", "dqp", "pcm", "bypass", "cip", "strong_intra", "tmvp", "wp",
               "rplm", "lt_ref", "scaling", "wpp", "tile_cols", "tile_rows", "slice_ctus", "dep_slices", "merge_cand", "cabac_init", "par_mrg", "rps_sps",
               "cb_qp_off", "cr_qp_off", "search", "open_gop")


class HevcGenParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in HEVC_FIELDS]


_hgen = None


def generate_hevc(recon_path=None, **kw):
    """HEVC Annex-B stream as bytes (tools/hevcgen.c).  Defaults: 176x144, 8 frames, QP 32, CTB 64, SAO + deblocking + TMVP + AMP on."""
    global _hgen
    if _hgen is None:
        p = os.path.join(_ROOT, "tools", "_build", "libhevcgen.so")
        if not os.path.exists(p):
            build_tools()
        _hgen = C.CDLL(p)
        _hgen.hevcgen_generate.argtypes = [C.POINTER(HevcGenParams), C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t), C.c_char_p]
        _hgen.hevcgen_free.argtypes = [C.c_void_p]
    d = dict(width=176, height=144, frames=8, qp=32, seed=0x4A4D0300, intra_period=32, deblock=1, sao=1, tmvp=1, amp=1, strong_intra=1, depth_inter=2, depth_intra=2)
    d.update(kw)
    unknown = set(d) - set(HEVC_FIELDS)
    if unknown:
        raise TypeError(f"generate_hevc: unknown parameters {sorted(unknown)}")
    p = HevcGenParams(*[int(d.get(n, 0)) for n in HEVC_FIELDS])
    buf = C.POINTER(C.c_ubyte)()
    n = C.c_size_t(0)
    if _hgen.hevcgen_generate(C.byref(p), C.byref(buf), C.byref(n), recon_path.encode() if recon_path else None) != 0:
        raise ValueError("hevcgen: bad parameters")
    out = C.string_at(buf, n.value)
    _hgen.hevcgen_free(buf)
    return out


def config_c3(frames=120, width=3840, height=2160, stream_id=0):
    """HEVC Main 4K60, 64x64 CTU, min CU 8, SAO + deblocking, random-access GOP 8, QP 32."""
    return dict(width=width, height=height, frames=frames, qp=32, seed=0x4A4D0000 + 3 * 256 + stream_id, intra_period=32, gop=8, num_ref=2,
                ctb_log2=6, min_cb_log2=3, sao=1, deblock=1, tmvp=1, amp=1, strong_intra=1, depth_inter=2, depth_intra=2, sdh=1)


class OracleHevc:
    """ctypes binding of oracle/_build/liborc_hevc.so (CPU oracle -- checker only)."""

    def __init__(self):
        p = os.path.join(_ROOT, "oracle", "_build", "liborc_hevc.so")
        if not os.path.exists(p):
            build_tools()
        L = C.CDLL(p)
        L.orch_decode_stream_to_buffer.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orch_free.argtypes = [C.c_void_p]
        L.orch_open.restype = C.c_void_p
        L.orch_open.argtypes = [C.c_void_p, C.c_void_p]
        L.orch_close.argtypes = [C.c_void_p]
        L.orch_decode_annexb.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orch_flush.argtypes = [C.c_void_p]
        L.orch_digest_enable.argtypes = [C.c_void_p]
        L.orch_digest_value.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.orch_digest_value.restype = C.c_uint64
        L.orch_last_error.argtypes = [C.c_void_p]
        L.orch_last_error.restype = C.c_char_p
        L.orch_tool_name.restype = C.c_char_p
        L.orch_tool_name.argtypes = [C.c_int]
        L.orch_tool_count.restype = C.c_long
        L.orch_tool_count.argtypes = [C.c_void_p, C.c_int]
        self.L = L

    def decode(self, data, out_fmt=1):
        buf = C.POINTER(C.c_ubyte)()
        n = C.c_size_t(0)
        w, h = C.c_int(0), C.c_int(0)
        cnt = self.L.orch_decode_stream_to_buffer(data, len(data), out_fmt, C.byref(buf), C.byref(n), C.byref(w), C.byref(h))
        if cnt < 0:
            raise RuntimeError("HEVC oracle decode failed")
        out = C.string_at(buf, n.value) if n.value else b""
        self.L.orch_free(buf)
        return out, cnt, w.value, h.value

    def _run(self, data, digest=False):
        d = self.L.orch_open(None, None)
        if digest:
            self.L.orch_digest_enable(d)
        rc = self.L.orch_decode_annexb(d, data, len(data))
        err = self.L.orch_last_error(d).decode()
        self.L.orch_flush(d)
        return d, rc, err

    def syntax_digest(self, data):
        d, rc, err = self._run(data, True)
        n = C.c_uint64(0)
        v = self.L.orch_digest_value(d, C.byref(n))
        self.L.orch_close(d)
        if rc < 0:
            raise RuntimeError("HEVC oracle: " + err)
        return v, n.value

    def tools(self, data):
        d, rc, err = self._run(data)
        out, i = {}, 0
        while True:
            name = self.L.orch_tool_name(i)
            if name is None:
                break
            c = self.L.orch_tool_count(d, i)
            if c:
                out[name.decode()] = c
            i += 1
        self.L.orch_close(d)
        if rc < 0:
            raise RuntimeError("HEVC oracle: " + err)
        return out
