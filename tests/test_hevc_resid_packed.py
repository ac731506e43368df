"""The packed inverse transform of k_hevc_resid / k_hevc_iresid (jmcodec_amd/csrc/hevc_resid_packed.h: coefficients scattered into row pairs, both stages as
v_dot2_i32_i16 sums against a pair table of the matrices) against a literal restatement of ITU-T H.265 8.6.4.2 in numpy, on the CPU (the header restates the
instruction in plain C++ for host builds; the check runs the stages task by task as the wave does).  The 4-point DST and the defining properties of the DCT
matrix (6.x of the table: first row 64, columns of the n-point matrices are the even rows of the 2n-point one) are checked against typed values."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = [[29, 55, 74, 84], [74, 74, 0, -74], [84, -29, -74, 55], [55, -84, 74, -29]]      # equation 8-x of 8.6.4.2 (transMatrix for nTbS = 4, DST-VII)
DCT4 = [[64, 64, 64, 64], [83, 36, -36, -83], [64, -64, -64, 64], [36, -83, 83, -36]]


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libhevc_resid_packed_check.so")
    src = os.path.join(ROOT, "tests", "native", "hevc_resid_packed_check.cpp")
    hdrs = [os.path.join(ROOT, "jmcodec_amd", "csrc", h) for h in ("hevc_resid_packed.h", "hevc_mc_packed.h", "mc_packed.h", "hevc_tables.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-unused-variable", "-Wno-unused-function", "-o", so, src])
    l = ctypes.CDLL(so)
    l.hrpc_block.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return l


def matrices(lib):
    t32 = np.array([[lib.hrpc_trans(j, y) for y in range(32)] for j in range(32)], dtype=np.int64)
    return t32, np.array([[lib.hrpc_dst(j, y) for y in range(4)] for j in range(4)], dtype=np.int64)


def test_matrices(lib):
    t32, dst = matrices(lib)
    assert dst.tolist() == DST
    assert t32[::8, :4].tolist() == DCT4                       # the 4-point matrix: rows 0, 8, 16, 24, first four columns
    assert (t32[0] == 64).all()
    for j in range(32):                                        # even / odd symmetry of row j (8.6.4.2: the columns mirror with sign (-1)^j)
        assert (t32[j, ::-1] == (t32[j] if j % 2 == 0 else -t32[j])).all()


def literal(d, M):
    """8.6.4.2 for 8-bit video: e = M^T d per column, g = Clip3(-32768, 32767, (e + 64) >> 7), r = (g M + 2048) >> 12 per row.  d[j][x], M[j][y]."""
    e = M.T @ d                                                # e[y][x] = sum_j M[j][y] d[j][x]
    g = np.clip((e + 64) >> 7, -32768, 32767)
    return (g @ M + 2048) >> 12                                # r[y][x] = sum_k g[y][k] M[k][x]


@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_inverse_transform_random_sparse_and_dense(lib, log2):
    t32, dst = matrices(lib)
    n = 1 << log2
    M = t32[::32 // n, :n]
    rng = np.random.default_rng(log2)
    res = (ctypes.c_int * (n * n))()
    cases = 0
    for trial in range(300):
        kind = trial % 6
        d = np.zeros((n, n), dtype=np.int64)
        if kind == 0:                                          # a handful of low-frequency coefficients (the common case)
            for _ in range(rng.integers(1, 6)):
                d[rng.integers(0, min(n, 4)), rng.integers(0, min(n, 4))] = rng.integers(-600, 600)
        elif kind == 1:                                        # DC only
            d[0, 0] = rng.integers(-32768, 32768)
        elif kind == 2:                                        # anywhere in the block, odd extents
            for _ in range(rng.integers(1, 12)):
                d[rng.integers(0, n), rng.integers(0, n)] = rng.integers(-2000, 2000)
        elif kind == 3:                                        # dense, small
            d = rng.integers(-64, 64, size=(n, n))
        elif kind == 4:                                        # extremes: the first stage must clip
            d = rng.choice([-32768, 32767, 0], size=(n, n), p=[0.2, 0.2, 0.6])
        else:                                                  # a single coefficient in the last row / column
            d[n - 1, rng.integers(0, n)] = rng.integers(-32768, 32768); d[rng.integers(0, n), n - 1] = rng.integers(-32768, 32768)
        pos = [(j, x) for j in range(n) for x in range(n) if d[j, x] != 0]
        if not pos:
            continue
        order = rng.permutation(len(pos))                      # the list is in coding (scan) order, not raster order
        coefs = np.array([(pos[i][0] * n + pos[i][1]) | ((int(d[pos[i]]) & 0xffff) << 16) for i in order], dtype=np.uint32)
        lib.hrpc_block(coefs.ctypes.data, len(coefs), log2, 0, res)
        want = literal(d, M)
        want16 = ((want + 32768) % 65536) - 32768              # the residual is kept in 16 bits (it always fits for conforming input; kind 4 may wrap)
        assert (np.array(res[:]).reshape(n, n) == want16).all(), (log2, trial, kind)
        cases += 1
    assert cases > 250


def test_dst_4x4(lib):
    t32, dst = matrices(lib)
    rng = np.random.default_rng(9)
    res = (ctypes.c_int * 16)()
    for _ in range(400):
        d = np.zeros((4, 4), dtype=np.int64)
        for _ in range(rng.integers(1, 10)):
            d[rng.integers(0, 4), rng.integers(0, 4)] = rng.integers(-32768, 32768)
        pos = [(j, x) for j in range(4) for x in range(4) if d[j, x] != 0]
        if not pos:
            continue
        coefs = np.array([(j * 4 + x) | ((int(d[j, x]) & 0xffff) << 16) for j, x in pos], dtype=np.uint32)
        lib.hrpc_block(coefs.ctypes.data, len(coefs), 2, 1, res)
        want = literal(d, dst)
        assert (np.array(res[:]).reshape(4, 4) == ((want + 32768) % 65536) - 32768).all()
