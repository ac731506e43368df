"""Seeded synthetic Annex-B streams (tools/h264gen.c, tools/hevcgen.c): INPUT generator for tests and bench.

Not part of the decode product: the reference ships no streams.  ``Oracle`` / ``OracleHevc`` are
re-exported from oracle/binding.py for the checker legs of tests/, smoke() and bench.py.
"""
import ctypes as C
import os
import subprocess
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from oracle.binding import Oracle, OracleHevc  # noqa: E402,F401  (checker only)


class GenParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "width", "height", "frames", "qp", "gop", "seed", "mode", "deblock", "num_ref", "slices",
        "pcm_only", "poc_type", "nonref_period", "alpha_off", "beta_off", "chroma_qp_off", "level_idc",
        "cip", "search", "cabac", "cabac_idc", "t8x8", "bframes", "direct_temporal", "wp", "dinf8", "scaling", "rplm", "mmco", "nc_corner", "no_intra",
        "fmo0", "poc_bottom", "paff", "gaps", "redundant", "vui_fps")]


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
             no_intra=0, fmo0=0, poc_bottom=0, paff=0, gaps=0, redundant=0, vui_fps=0,
             recon_path=None):
    """Returns the Annex-B stream as bytes (optionally writing the encoder's own reconstruction)."""
    p = GenParams(width, height, frames, qp, gop, seed, mode, deblock, num_ref, slices, pcm_only, poc_type,
                  nonref_period, alpha_off, beta_off, chroma_qp_off, level_idc, cip, search, cabac, cabac_idc, t8x8,
                  bframes, direct_temporal, wp, dinf8, scaling, rplm, mmco, nc_corner, no_intra, fmo0, poc_bottom, paff, gaps, redundant, vui_fps)
    buf = C.POINTER(C.c_ubyte)()
    n = C.c_size_t(0)
    rc = _genlib().h264gen_generate(C.byref(p), C.byref(buf), C.byref(n), recon_path.encode() if recon_path else None)
    if rc != 0:
        raise ValueError("h264gen: bad parameters")
    data = C.string_at(buf, n.value)
    _genlib().h264gen_free(buf)
    return data


def last_pocs():
    """PicOrderCnt per display index as the generator meant it, of the H.264 stream THIS thread generated last (tools/h264gen.c h264gen_last_pocs)."""
    buf = (C.c_int * 4096)()
    n = _genlib().h264gen_last_pocs(buf, 4096)
    return list(buf[:min(n, 4096)])


# BASELINE.json configs restated as generator parameters; seed = 0x4A4D0000 + config*256 + stream
def config_c1(stream_id=0, frames=300, width=1920, height=1080):
    return dict(width=width, height=height, frames=frames, qp=28, gop=30, seed=0x4A4D0000 + 1 * 256 + stream_id,
                mode=0, deblock=1, num_ref=1, level_idc=40)


def config_c2(stream_id=0, frames=120, width=3840, height=2160):
    """BASELINE config 2: H.264 High 4K, CABAC, 8x8 transform, I B B P with two references."""
    return dict(width=width, height=height, frames=frames, qp=30, gop=30, seed=0x4A4D0000 + 2 * 256 + stream_id,
                mode=0, deblock=1, num_ref=2, level_idc=52, cabac=1, t8x8=1, bframes=2, poc_type=0)


# ---------------------------------------------------------------------------------------------------------
# HEVC: tools/hevcgen.c (stream generator) and oracle/orc_hevc_*.c (CPU oracle) -- test infrastructure
# ---------------------------------------------------------------------------------------------------------
HEVC_FIELDS = ("width", "height", "frames", "qp", "seed", "intra_period", "gop", "num_ref", "ctb_log2", "min_cb_log2", "max_tb_log2", "min_tb_log2",
               "depth_inter", "depth_intra", "mode", "amp", "sao", "deblock", "tskip", "sdh", "dqp", "pcm", "bypass", "cip", "strong_intra", "tmvp", "wp",
               "rplm", "lt_ref", "scaling", "wpp", "tile_cols", "tile_rows", "slice_ctus", "dep_slices", "merge_cand", "cabac_init", "par_mrg", "rps_sps",
               "cb_qp_off", "cr_qp_off", "search", "open_gop", "vui_fps")


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
    d = dict(width=176, height=144, frames=8, qp=32, seed=0x4A4D0300, intra_period=32, deblock=1, sao=1, tmvp=1, amp=1, strong_intra=1, depth_inter=2,
        depth_intra=2)
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


def idr_period(data, which=0, is_hevc=False):
    """IDR period `which` of the stream as a stream of its own: the parameter sets in front of the first picture, then the access units from the
    period's IDR picture up to (not including) the next one.  An IDR picture resets the decoding process, so period k decodes to frames
    k*period .. of the full stream's display order (the generator's GOPs are closed)."""
    nal_starts = [k for k in range(len(data) - 3) if data[k] == 0 and data[k + 1] == 0 and data[k + 2] == 1]
    ntype = lambda k: ((data[k + 3] >> 1) & 63) if is_hevc else (data[k + 3] & 31)
    is_ps = lambda t: (t in (32, 33, 34, 35, 39)) if is_hevc else (t in (6, 7, 8, 9))
    idr = []
    for k in nal_starts:
        t = ntype(k)
        if (t in (19, 20)) if is_hevc else (t == 5):
            first_slice = (data[k + 5] & 0x80) != 0 if is_hevc else (data[k + 4] & 0x80) != 0     # first_slice_segment_in_pic_flag / first_mb_in_slice == 0
            if first_slice:
                idr.append(k)

    def cut_before(pos):
        # parameter sets that precede an IDR picture belong to it: cut before them (they follow the previous picture's last slice)
        prev = [k for k in nal_starts if k < pos]
        while prev and is_ps(ntype(prev[-1])):
            pos = prev.pop()
        while pos > 0 and data[pos - 1] == 0:
            pos -= 1
        return pos
    if which >= len(idr):
        return None
    end = cut_before(idr[which + 1]) if which + 1 < len(idr) else len(data)
    if which == 0:
        return data[:end]
    head = data[:next((k for k in nal_starts if not is_ps(ntype(k))), 0)]     # the parameter sets in front of the first picture
    while head and head[-1:] == b"\x00":
        head = head[:-1]
    return head + data[cut_before(idr[which]):end]
