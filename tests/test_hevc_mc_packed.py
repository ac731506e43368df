"""The packed interpolation arithmetic of k_hevc_mc (jmcodec_amd/csrc/hevc_mc_packed.h: v_dot4 horizontal sums on signed bytes, 16-bit row pairs stored
column-major, v_dot2 vertical sums) against a literal restatement of ITU-T H.265 8.5.3.3.3.1 (luma) / 8.5.3.3.3.2 (chroma) and the default weighted
prediction of 8.5.3.3.4.2, on the CPU: the header restates the GPU instructions it uses in plain C++ for host builds and the check runs the two passes lane
by lane as the wave does.  The taps are typed here a second time from Tables 8-11 / 8-12."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Table 8-11: fL[p][0..7] for quarter-sample positions 1..3; Table 8-12: fC[p][0..3] for eighth-sample positions 1..7
FL = {1: [-1, 4, -10, 58, 17, -5, 1, 0], 2: [-1, 4, -11, 40, 40, -11, 4, -1], 3: [0, 1, -5, 17, 58, -10, 4, -1]}
FC = {1: [-2, 58, 10, -2], 2: [-4, 54, 16, -2], 3: [-6, 46, 28, -4], 4: [-4, 36, 36, -4], 5: [-4, 28, 46, -6], 6: [-2, 16, 54, -4], 7: [-2, 10, 58, -2]}


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libhevc_mc_packed_check.so")
    src = os.path.join(ROOT, "tests", "native", "hevc_mc_packed_check.cpp")
    hdrs = [os.path.join(ROOT, "jmcodec_amd", "csrc", h) for h in ("hevc_mc_packed.h", "mc_packed.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", so, src])
    l = ctypes.CDLL(so)
    l.hmcp_weigh_default4.restype = ctypes.c_uint32
    l.hmcp_block.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p]
    return l


def test_taps_are_the_tables(lib):
    for f, t in FL.items():
        assert [lib.hmcp_luma_tap(f, i) for i in range(8)] == t and sum(t) == 64
    for f, t in FC.items():
        assert [lib.hmcp_chroma_tap(f, i) for i in range(4)] == t and sum(t) == 64
    assert [lib.hmcp_luma_tap(0, i) for i in range(8)] == [0, 0, 0, 1, 0, 0, 0, 0]
    assert [lib.hmcp_chroma_tap(0, i) for i in range(4)] == [0, 1, 0, 0]


def luma_literal(ref, x, y, xf, yf):
    """8.5.3.3.3.1 for 8-bit samples (shift1 = 0, shift2 = 6, shift3 = 6): predSampleLX of the sample whose integer position is (x, y)."""
    A = lambda dx, dy: int(ref[y + dy][x + dx])
    if xf == 0 and yf == 0:
        return A(0, 0) << 6
    if yf == 0:
        return sum(FL[xf][i] * A(i - 3, 0) for i in range(8))
    if xf == 0:
        return sum(FL[yf][i] * A(0, i - 3) for i in range(8))
    col = [sum(FL[xf][i] * A(i - 3, j - 3) for i in range(8)) for j in range(8)]      # a / b / c of the eight rows (shift1 = 0)
    return sum(FL[yf][j] * col[j] for j in range(8)) >> 6


def chroma_literal(ref, x, y, xf, yf):
    """8.5.3.3.3.2 on one component plane."""
    B = lambda dx, dy: int(ref[y + dy][x + dx])
    if xf == 0 and yf == 0:
        return B(0, 0) << 6
    if yf == 0:
        return sum(FC[xf][i] * B(i - 1, 0) for i in range(4))
    if xf == 0:
        return sum(FC[yf][i] * B(0, i - 1) for i in range(4))
    col = [sum(FC[xf][i] * B(i - 1, j - 1) for i in range(4)) for j in range(4)]
    return sum(FC[yf][j] * col[j] for j in range(4)) >> 6


@pytest.mark.parametrize("bw,bh", [(16, 16), (8, 8), (4, 8), (8, 4), (16, 12), (12, 16), (16, 4), (4, 16)])
def test_luma_every_fraction_every_alignment(lib, bw, bh):
    rng = np.random.default_rng(bw * 100 + bh)
    ref = rng.integers(0, 256, size=(64, 128), dtype=np.uint8)
    ref[:8, :] = 255                     # saturated rows: the extremes of the 16-bit intermediates
    ref[8:16, :] = 0
    out = (ctypes.c_int * (bw * bh))()
    for xf in range(4):
        for yf in range(4):
            for xi, yi in ((20, 3), (21, 9), (22, 20), (23, 30), (40, 4)):      # every dword alignment of the window, odd and even rows
                lib.hmcp_block(ref.ctypes.data, ref.strides[0], xi - 3, yi - 3, bw, bh, xf, yf, 0, out)
                got = np.array(out[:], dtype=np.int64).reshape(bh, bw)
                want = np.array([[luma_literal(ref, xi + c, yi + r, xf, yf) for c in range(bw)] for r in range(bh)])
                assert (got == want).all(), (bw, bh, xf, yf, xi, yi)


@pytest.mark.parametrize("bw,bh", [(8, 8), (4, 4), (2, 4), (4, 2), (8, 6), (6, 8), (8, 2), (2, 8)])
def test_chroma_every_fraction_every_alignment(lib, bw, bh):
    """bw x bh CbCr pairs of an interleaved plane."""
    rng = np.random.default_rng(bw * 10 + bh)
    cb = rng.integers(0, 256, size=(48, 64), dtype=np.uint8)
    cr = rng.integers(0, 256, size=(48, 64), dtype=np.uint8)
    cb[:6, :] = 255; cr[:6, :] = 0
    plane = np.empty((48, 128), dtype=np.uint8)
    plane[:, 0::2] = cb; plane[:, 1::2] = cr
    out = (ctypes.c_int * (2 * bw * bh))()
    for xf in range(8):
        for yf in range(8):
            for xi, yi in ((10, 2), (11, 7), (12, 12), (13, 21)):
                lib.hmcp_block(plane.ctypes.data, plane.strides[0], 2 * (xi - 1), yi - 1, bw, bh, xf, yf, 1, out)
                got = np.array(out[:], dtype=np.int64).reshape(bh, 2 * bw)
                for comp, pl in ((0, cb), (1, cr)):
                    want = np.array([[chroma_literal(pl, xi + c, yi + r, xf, yf) for c in range(bw)] for r in range(bh)])
                    assert (got[:, comp::2] == want).all(), (bw, bh, xf, yf, xi, yi, comp)


def test_default_weighted_prediction(lib):
    """8.5.3.3.4.2: shift1 = 6, uni-prediction Clip1((predSamples + 32) >> 6); bi-prediction Clip1((a + b + 64) >> 7)."""
    rng = np.random.default_rng(5)
    for _ in range(2000):
        a = [int(v) for v in rng.integers(-1500, 17900, size=4)]         # the range of 14-bit intermediates of 8-bit video (with filter overshoot)
        b = [int(v) for v in rng.integers(-1500, 17900, size=4)]
        A, B = (ctypes.c_int * 4)(*a), (ctypes.c_int * 4)(*b)
        uni = lib.hmcp_weigh_default4(A, B, 0)
        bi = lib.hmcp_weigh_default4(A, B, 1)
        clip = lambda v: max(0, min(255, v))
        assert [(uni >> (8 * k)) & 255 for k in range(4)] == [clip((a[k] + 32) >> 6) for k in range(4)]
        assert [(bi >> (8 * k)) & 255 for k in range(4)] == [clip((a[k] + b[k] + 64) >> 7) for k in range(4)]
