"""The packed sample arithmetic of the inter reconstruction (jmcodec_amd/csrc/mc_packed.h: v_dot4 six-tap sums, v_perm transposes, v_lerp_u8 averages,
v_sat_pk_u8_i16 clips) against a literal restatement of H.264 8.4.2.2.1 / 8.4.2.2.2, on the CPU: the header restates the seven GPU instructions it uses
in plain C++ for host builds.  The GPU parity tests run the same functions with the real instructions against the oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libmc_packed_check.so")
    src = os.path.join(ROOT, "tests", "native", "mc_packed_check.cpp")
    hdr = os.path.join(ROOT, "jmcodec_amd", "csrc", "mc_packed.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", so, src])
    l = ctypes.CDLL(so)
    l.mcp_luma4.restype = ctypes.c_uint32
    l.mcp_luma4.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5
    for f in (l.mcp_chroma_uv, l.mcp_add_residual4, l.mcp_add_residual_uv, l.mcp_lerp):
        f.restype = ctypes.c_uint32
    return l


def clip1(v):
    return max(0, min(255, v))


def tap6(a, b, c, d, e, f):
    return a - 5 * b + 20 * c + 20 * d - 5 * e + f


def luma_literal(s, x, y, fx, fy):
    """8.4.2.2.1, equations 8-241 .. 8-261, for the sample at integer position (x, y) of the 2-D array s (rows, columns)."""
    P = lambda dx, dy: int(s[y + dy][x + dx])
    b1 = lambda dy: tap6(P(-2, dy), P(-1, dy), P(0, dy), P(1, dy), P(2, dy), P(3, dy))
    h1 = lambda dx: tap6(P(dx, -2), P(dx, -1), P(dx, 0), P(dx, 1), P(dx, 2), P(dx, 3))
    G, H, M = P(0, 0), P(1, 0), P(0, 1)
    b, h = clip1((b1(0) + 16) >> 5), clip1((h1(0) + 16) >> 5)
    s_, m = clip1((b1(1) + 16) >> 5), clip1((h1(1) + 16) >> 5)
    j = clip1((tap6(b1(-2), b1(-1), b1(0), b1(1), b1(2), b1(3)) + 512) >> 10)
    table = {(0, 0): G, (1, 0): (G + b + 1) >> 1, (2, 0): b, (3, 0): (H + b + 1) >> 1,
             (0, 1): (G + h + 1) >> 1, (1, 1): (b + h + 1) >> 1, (2, 1): (b + j + 1) >> 1, (3, 1): (b + m + 1) >> 1,
             (0, 2): h, (1, 2): (h + j + 1) >> 1, (2, 2): j, (3, 2): (j + m + 1) >> 1,
             (0, 3): (M + h + 1) >> 1, (1, 3): (h + s_ + 1) >> 1, (2, 3): (j + s_ + 1) >> 1, (3, 3): (m + s_ + 1) >> 1}
    return table[(fx, fy)]


def make_window(samples, stride_dw):
    """samples: (rows, cols) uint8 -> the LDS image: samples ^ 0x80, stride_dw dwords per row (+ one spare row of padding)"""
    rows, cols = samples.shape
    img = np.zeros((rows + 2, stride_dw * 4), dtype=np.uint8)
    img[:rows, :cols] = samples ^ 0x80
    return np.ascontiguousarray(img).view(np.uint32).reshape(-1)


@pytest.mark.parametrize("kind", ["random", "extremes", "checker"])
def test_luma_every_fractional_position(lib, kind):
    rng = np.random.default_rng(0x4A4D0500 + len(kind))
    rows, cols = 21, 24
    for trial in range(6):
        if kind == "random":
            s = rng.integers(0, 256, size=(rows, cols), dtype=np.uint8)
        elif kind == "extremes":
            s = rng.choice(np.array([0, 255], dtype=np.uint8), size=(rows, cols))
        else:
            s = (((np.add.outer(np.arange(rows), np.arange(cols)) + trial) & 1) * 255).astype(np.uint8)
        for stride in (5, 6):
            win = make_window(s[:, :stride * 4], stride)
            ptr = win.ctypes.data
            for fy in range(4):
                for fx in range(4):
                    for wr in (0, 3, rows - 6):
                        for cb in range(0, stride * 4 - 11):
                            got = lib.mcp_luma4(ptr, stride, wr, cb, fx, fy)
                            want = [luma_literal(s, cb + 2 + k, wr + 2, fx, fy) for k in range(4)]
                            assert [(got >> (8 * k)) & 255 for k in range(4)] == want, (kind, trial, stride, fx, fy, wr, cb)


def test_chroma_every_fractional_position(lib):
    rng = np.random.default_rng(0x4A4D0501)
    for trial in range(200):
        n = rng.integers(0, 256, size=8) if trial > 8 else np.array([255] * 8 if trial & 1 else [0, 255, 255, 0, 255, 0, 0, 255])
        ua, va, ub, vb, uc, vc, ud, vd = (int(v) for v in n)
        wa = ua | va << 8 | ub << 16 | vb << 24
        wb = uc | vc << 8 | ud << 16 | vd << 24
        for fy in range(8):
            for fx in range(8):
                got = lib.mcp_chroma_uv(wa, wb, fx, fy)
                u = ((8 - fx) * (8 - fy) * ua + fx * (8 - fy) * ub + (8 - fx) * fy * uc + fx * fy * ud + 32) >> 6      # 8-270
                v = ((8 - fx) * (8 - fy) * va + fx * (8 - fy) * vb + (8 - fx) * fy * vc + fx * fy * vd + 32) >> 6
                assert got == (u | v << 8), (trial, fx, fy)


def test_residual_add_clips_like_clip1(lib):
    rng = np.random.default_rng(0x4A4D0502)
    cases = [((0, 255, 128, 1), (-1, 1, -32768, 32767)), ((255, 255, 0, 0), (32767, 1, -1, -32768))]
    for _ in range(500):
        cases.append((tuple(int(v) for v in rng.integers(0, 256, size=4)),
                      tuple(int(v) for v in rng.choice([rng.integers(-600, 600), rng.integers(-32768, 32768)], size=4))))
    for p, r in cases:
        pred = p[0] | p[1] << 8 | p[2] << 16 | p[3] << 24
        got = lib.mcp_add_residual4(pred, *r)
        assert [(got >> (8 * k)) & 255 for k in range(4)] == [clip1(p[k] + r[k]) for k in range(4)], (p, r)
        got = lib.mcp_add_residual_uv(p[0] | p[1] << 8, r[0], r[1])
        assert got == (clip1(p[0] + r[0]) | clip1(p[1] + r[1]) << 8), (p, r)


def test_lerp_is_the_rounded_average(lib):
    rng = np.random.default_rng(0x4A4D0503)
    for _ in range(300):
        a, b = (int(v) for v in rng.integers(0, 2 ** 32, size=2, dtype=np.uint64))
        got = lib.mcp_lerp(a, b)
        assert [(got >> (8 * k)) & 255 for k in range(4)] == [((((a >> (8 * k)) & 255) + ((b >> (8 * k)) & 255) + 1) >> 1) for k in range(4)]


def test_macroblock_row_by_multiplication():
    """k_recon_inter takes the macroblock row as mulhi(address, ceil(2^32 / mb_w)) (PicParams.mb_w_magic, decoder.cpp; 0 stands for mb_w == 1).  The parser
    admits pictures of up to 1024 x 1024 macroblocks (h264_syntax.cpp); the identity must hold for every address of every such picture."""
    for mb_w in range(2, 1025):
        magic = ((1 << 32) + mb_w - 1) // mb_w
        assert magic < (1 << 32)
        n = np.arange(0, mb_w * 1024, dtype=np.uint64)
        # the addresses of whole pictures: first / last of each row are the critical ones, but all are cheap enough at a stride
        step = 1 if mb_w < 64 else 7
        n = np.concatenate([n[::step], np.arange(mb_w - 1, mb_w * 1024, mb_w, dtype=np.uint64), np.arange(0, mb_w * 1024, mb_w, dtype=np.uint64)])
        assert np.array_equal((n * np.uint64(magic)) >> np.uint64(32), n // np.uint64(mb_w)), mb_w
    assert (((1 << 32) + 1 - 1) // 1) & 0xffffffff == 0                 # mb_w == 1: the 32-bit field reads 0, the kernel then takes row = address
