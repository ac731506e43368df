"""Chain launches fetch ONE reference window for the four macroblocks of a workgroup when their one-window fetches lie close together (recon_device.h, the
"quad" path; geometry in mc_packed.h: quad_geometry).  On the CPU: the four macroblocks predicted out of the shared window must get exactly the samples they
get out of their private windows -- every fractional position, vectors from identical to as far apart as the geometry accepts, windows at the picture's
edges -- and nothing outside the loaded extent may matter (the unloaded part of the window holds noise that differs between two runs)."""
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
    so = os.path.join(out, "libquad_window_check.so")
    src = os.path.join(ROOT, "tests", "native", "quad_window_check.cpp")
    hdr = os.path.join(ROOT, "jmcodec_amd", "csrc", "mc_packed.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", so, src])
    return ctypes.CDLL(so)


W, H, PITCH = 256, 96, 256


def picture(rng):
    luma = rng.integers(0, 256, (H, PITCH), dtype=np.uint8)
    chroma = rng.integers(0, 256, (H // 2 + 2, PITCH), dtype=np.uint8)       # (two spare rows: the loads are whole dwords)
    return np.ascontiguousarray(np.concatenate([luma, chroma])), H * PITCH


def run(lib, pic, chroma_offset, votes, frac, noise):
    a = np.zeros(4 * 384, dtype=np.uint8)
    b = np.zeros(4 * 384, dtype=np.uint8)
    v = np.ascontiguousarray(np.array(votes, dtype=np.int32).reshape(-1))
    f = np.ascontiguousarray(np.array(frac, dtype=np.int32).reshape(-1))
    ok = lib.qw_check(pic.ctypes.data_as(ctypes.c_void_p), PITCH, chroma_offset, v.ctypes.data_as(ctypes.c_void_p), f.ctypes.data_as(ctypes.c_void_p),
                      ctypes.c_uint32(noise), a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p))
    return ok, a, b


def votes_for(mbx0, mby, mvs):
    """the one-window parameters of macroblocks mbx0 .. mbx0 + 3 of row mby with quarter-sample vectors mvs (recon_device.h: xi, yi, 2 * cxi, cyi)"""
    out = []
    for w, (mvx, mvy) in enumerate(mvs):
        mbx = mbx0 + w
        out.append([0, mbx * 16 + (mvx >> 2) - 2, mby * 16 + (mvy >> 2) - 2, 2 * (mbx * 8 + (mvx >> 3)), mby * 8 + (mvy >> 3)])
    return out


def inside(v):
    return all(0 <= xi <= W - 21 and 0 <= yi <= H - 21 for _, xi, yi, _, _ in v)


def test_shared_window_equals_private_windows(lib):
    rng = np.random.default_rng(0x51AD)
    pic, co = picture(rng)
    accepted = rejected = 0
    for trial in range(1500):
        mbx0 = int(rng.integers(0, (W // 16) - 3)) & ~3 if trial % 3 else 4
        mby = int(rng.integers(0, H // 16))
        spread = [0, 0, 2, 6, 12, 30, 60][trial % 7]
        base = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
        mvs = [(base[0] + int(rng.integers(-spread, spread + 1)), base[1] + int(rng.integers(-spread, spread + 1))) for _ in range(4)]
        v = votes_for(mbx0, mby, mvs)
        if not inside(v):
            continue
        frac = [[mx & 3, my & 3, mx & 7, my & 7] for mx, my in mvs]
        ok1, a1, b1 = run(lib, pic, co, v, frac, 0x9E3779B1)
        ok2, a2, b2 = run(lib, pic, co, v, frac, 0x7F4A7C15)
        assert ok1 == ok2
        if not ok1:
            rejected += 1
            continue
        accepted += 1
        assert (a1 == b1).all(), (trial, v, frac)
        assert (a1 == a2).all() and (b1 == b2).all(), ("a sample depends on what lies outside the loaded extent", trial, v)
    assert accepted > 300 and rejected > 20, (accepted, rejected)


def test_geometry_rejects_what_does_not_fit(lib):
    rng = np.random.default_rng(7)
    pic, co = picture(rng)
    base = votes_for(4, 2, [(0, 0)] * 4)
    assert run(lib, pic, co, base, [[0, 0, 0, 0]] * 4, 1)[0] == 1
    other = [list(r) for r in base]; other[2][0] = 1                      # another reference picture
    assert run(lib, pic, co, other, [[0, 0, 0, 0]] * 4, 1)[0] == 0
    none = [list(r) for r in base]; none[1][0] = -1                       # a macroblock that does not take the one-window path
    assert run(lib, pic, co, none, [[0, 0, 0, 0]] * 4, 1)[0] == 0
    wide = votes_for(4, 2, [(-80, 0), (0, 0), (0, 0), (80, 0)])           # 40 samples apart: a row of the four no longer fits 27 dwords
    assert run(lib, pic, co, wide, [[0, 0, 0, 0]] * 4, 1)[0] == 0
    tall = votes_for(4, 2, [(0, -32), (0, 0), (0, 0), (0, 32)])           # 16 rows apart: more than 32 rows
    assert run(lib, pic, co, tall, [[0, 0, 0, 0]] * 4, 1)[0] == 0
