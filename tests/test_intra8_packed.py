"""Intra8x8 prediction on packed bytes (jmcodec_amd/csrc/intra8_packed.h: the reference path of a block in seven registers, the [1 2 1] filter as two
v_lerp_u8, the taps through v_perm_b32 selectors from a table) against a literal restatement of H.264 8.3.2.2 on the CPU -- every mode, every sample, the
availability patterns a picture can produce, the extreme sample values.  The GPU parity tests run the same functions with the real instructions against the
oracle (High-profile cases of tests/test_gpu_parity.py)."""
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
    so = os.path.join(out, "libintra8_packed_check.so")
    src = os.path.join(ROOT, "tests", "native", "intra8_packed_check.cpp")
    hdrs = [os.path.join(ROOT, "jmcodec_amd", "csrc", h) for h in ("intra8_packed.h", "mc_packed.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", so, src])
    return ctypes.CDLL(so)


def reference_samples(left, corner, top, a, b, c, d):
    """8.3.2.2: the neighbouring samples p[x, y] as a dict, with the substitution for missing samples above right; samples of a neighbour that is not
    available count as 128 (the convention of this decoder and its oracle for non-conforming mode choices)."""
    p = {}
    for y in range(8):
        p[(-1, y)] = int(left[y]) if a else 128
    p[(-1, -1)] = int(corner) if d else 128
    for x in range(16):
        p[(x, -1)] = (int(top[x]) if (x < 8 or c) else int(top[7])) if b else 128
    return p


def filtered(p, a, b, d):
    """8.3.2.2.1 reference sample filtering, equations 8-78 .. 8-87"""
    f = {}
    f[(0, -1)] = (p[(-1, -1)] + 2 * p[(0, -1)] + p[(1, -1)] + 2) >> 2 if d else (3 * p[(0, -1)] + p[(1, -1)] + 2) >> 2
    for x in range(1, 15):
        f[(x, -1)] = (p[(x - 1, -1)] + 2 * p[(x, -1)] + p[(x + 1, -1)] + 2) >> 2
    f[(15, -1)] = (p[(14, -1)] + 3 * p[(15, -1)] + 2) >> 2
    if not d:
        f[(-1, -1)] = 128
    elif not a and not b:
        f[(-1, -1)] = p[(-1, -1)]
    elif not b:
        f[(-1, -1)] = (3 * p[(-1, -1)] + p[(-1, 0)] + 2) >> 2
    elif not a:
        f[(-1, -1)] = (3 * p[(-1, -1)] + p[(0, -1)] + 2) >> 2
    else:
        f[(-1, -1)] = (p[(0, -1)] + 2 * p[(-1, -1)] + p[(-1, 0)] + 2) >> 2
    f[(-1, 0)] = (p[(-1, -1)] + 2 * p[(-1, 0)] + p[(-1, 1)] + 2) >> 2 if d else (3 * p[(-1, 0)] + p[(-1, 1)] + 2) >> 2
    for y in range(1, 7):
        f[(-1, y)] = (p[(-1, y - 1)] + 2 * p[(-1, y)] + p[(-1, y + 1)] + 2) >> 2
    f[(-1, 7)] = (p[(-1, 6)] + 3 * p[(-1, 7)] + 2) >> 2
    return f


def predict_literal(f, mode, a, b):
    """8.3.2.2.2 .. 8.3.2.2.10 on the filtered samples f[(x, y)]"""
    P = lambda x, y: f[(x, y)]
    out = np.zeros((8, 8), dtype=np.int32)
    for y in range(8):
        for x in range(8):
            if mode == 0:
                v = P(x, -1)
            elif mode == 1:
                v = P(-1, y)
            elif mode == 2:
                st, sl = sum(P(i, -1) for i in range(8)), sum(P(-1, i) for i in range(8))
                v = (st + sl + 8) >> 4 if (a and b) else (sl + 4) >> 3 if a else (st + 4) >> 3 if b else 128
            elif mode == 3:
                v = (P(14, -1) + 3 * P(15, -1) + 2) >> 2 if x == 7 and y == 7 else (P(x + y, -1) + 2 * P(x + y + 1, -1) + P(x + y + 2, -1) + 2) >> 2
            elif mode == 4:
                if x > y:
                    v = (P(x - y - 2, -1) + 2 * P(x - y - 1, -1) + P(x - y, -1) + 2) >> 2
                elif x < y:
                    v = (P(-1, y - x - 2) + 2 * P(-1, y - x - 1) + P(-1, y - x) + 2) >> 2
                else:
                    v = (P(0, -1) + 2 * P(-1, -1) + P(-1, 0) + 2) >> 2
            elif mode == 5:
                z, i = 2 * x - y, x - (y >> 1)
                if z >= 0 and z % 2 == 0:
                    v = (P(i - 1, -1) + P(i, -1) + 1) >> 1
                elif z >= 0:
                    v = (P(i - 2, -1) + 2 * P(i - 1, -1) + P(i, -1) + 2) >> 2
                elif z == -1:
                    v = (P(-1, 0) + 2 * P(-1, -1) + P(0, -1) + 2) >> 2
                else:
                    v = (P(-1, y - 2 * x - 1) + 2 * P(-1, y - 2 * x - 2) + P(-1, y - 2 * x - 3) + 2) >> 2
            elif mode == 6:
                z, i = 2 * y - x, y - (x >> 1)
                if z >= 0 and z % 2 == 0:
                    v = (P(-1, i - 1) + P(-1, i) + 1) >> 1
                elif z >= 0:
                    v = (P(-1, i - 2) + 2 * P(-1, i - 1) + P(-1, i) + 2) >> 2
                elif z == -1:
                    v = (P(-1, 0) + 2 * P(-1, -1) + P(0, -1) + 2) >> 2
                else:
                    v = (P(x - 2 * y - 1, -1) + 2 * P(x - 2 * y - 2, -1) + P(x - 2 * y - 3, -1) + 2) >> 2
            elif mode == 7:
                i = x + (y >> 1)
                v = (P(i, -1) + P(i + 1, -1) + 1) >> 1 if y % 2 == 0 else (P(i, -1) + 2 * P(i + 1, -1) + P(i + 2, -1) + 2) >> 2
            else:
                z, i = x + 2 * y, y + (x >> 1)
                if z > 13:
                    v = P(-1, 7)
                elif z == 13:
                    v = (P(-1, 6) + 3 * P(-1, 7) + 2) >> 2
                elif z % 2 == 0:
                    v = (P(-1, i) + P(-1, i + 1) + 1) >> 1
                else:
                    v = (P(-1, i) + 2 * P(-1, i + 1) + P(-1, i + 2) + 2) >> 2
            out[y, x] = v
    return out


# (a, b, c, d): left, above, above right, above left.  Everything a picture can produce (slice groups included: a corner without the neighbours beside it);
# samples above right without samples above are never used (the substitution of 8.3.2.2 needs p[7, -1]), that combination is left out.
PATTERNS = [(a, b, c, d) for a in (0, 1) for b in (0, 1) for c in (0, 1) for d in (0, 1) if not (c and not b)]


def run_block(lib, left, corner, top, junk, a, b, c, d, mode):
    out = np.zeros(64, dtype=np.uint8)
    lib.i8p_block(left.ctypes.data_as(ctypes.c_void_p), int(corner), top.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(junk), a, b, c, d, mode,
                  out.ctypes.data_as(ctypes.c_void_p))
    return out.reshape(8, 8).astype(np.int32)


def test_filtered_path_matches_8_3_2_2_1(lib):
    rng = np.random.default_rng(0x1808)
    for trial in range(200):
        left = rng.integers(0, 256, 8, dtype=np.uint8)
        top = rng.integers(0, 256, 16, dtype=np.uint8)
        corner = int(rng.integers(0, 256))
        if trial % 7 == 0:
            left[:] = 255; top[:] = 255; corner = 255
        for a, b, c, d in PATTERNS:
            got = np.zeros(25, dtype=np.uint8)
            lib.i8p_path(left.ctypes.data_as(ctypes.c_void_p), corner, top.ctypes.data_as(ctypes.c_void_p), a, b, c, d, got.ctypes.data_as(ctypes.c_void_p))
            f = filtered(reference_samples(left, corner, top, a, b, c, d), a, b, d)
            want = [f[(-1, 7 - k)] for k in range(8)] + [f[(-1, -1)]] + [f[(x, -1)] for x in range(16)]
            assert list(got) == want, (trial, a, b, c, d)


@pytest.mark.parametrize("mode", range(9))
def test_every_mode_matches_8_3_2_2(lib, mode):
    rng = np.random.default_rng(0x1808 + mode)
    for trial in range(120):
        left = rng.integers(0, 256, 8, dtype=np.uint8)
        top = rng.integers(0, 256, 16, dtype=np.uint8)
        corner = int(rng.integers(0, 256))
        if trial == 0:
            left[:] = 255; top[:] = 255; corner = 255
        if trial == 1:
            left[:] = 0; top[:] = 0; corner = 0
        if trial == 2:
            left[:] = [0, 255] * 4; top[:] = [255, 0] * 8; corner = 255
        junk = int(rng.integers(0, 1 << 32))
        for a, b, c, d in PATTERNS:
            want = predict_literal(filtered(reference_samples(left, corner, top, a, b, c, d), a, b, d), mode, a, b)
            got = run_block(lib, left, corner, top, junk, a, b, c, d, mode)
            assert (got == want).all(), (mode, trial, a, b, c, d, got.tolist(), want.tolist())
