"""The order of reconstruction groups inside one key bucket of a chain launch's work list (jmcodec_amd/csrc/chain_order.h): a picture's 8-macroblock column
should always run on the same XCD pair (work-list position % 4).  Whatever the order, it must be a PERMUTATION of the bucket -- the key rule of the chain
launches (tools/chain_keys.py) orders buckets, so nothing may leave its bucket, be lost or be doubled -- and a position must get a group of its class whenever
the bucket still holds one."""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libchain_order_check.so")
    src = os.path.join(ROOT, "tests", "native", "chain_order_check.cpp")
    hdr = os.path.join(ROOT, "jmcodec_amd", "csrc", "chain_order.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", so, src])
    l = ctypes.CDLL(so)
    l.co_append.restype = ctypes.c_int
    return l


def entry(pic, row, col):
    return pic << 16 | row * 32 + col


def append(lib, bucket, out, pos):
    b = np.ascontiguousarray(np.array(bucket, dtype=np.uint32))
    return lib.co_append(b.ctypes.data_as(ctypes.c_void_p), len(bucket), out.ctypes.data_as(ctypes.c_void_p), pos)


def test_a_work_list_keeps_its_buckets_and_loses_nothing(lib):
    rng = random.Random(0xC4A1)
    for trial in range(200):
        buckets = []
        for k in range(rng.randrange(1, 40)):
            n = rng.choice([0, 1, 2, 3, 5, 8, 15, 30])
            buckets.append([entry(rng.randrange(0, 24), rng.randrange(0, 68), rng.randrange(0, 15)) for _ in range(n)])
        total = sum(len(b) for b in buckets)
        out = np.zeros(total + 16, dtype=np.uint32)
        pos = rng.randrange(0, 7)                   # (the bands of the launch come first: the groups do not start at position 0)
        start = pos
        bounds = []
        for b in buckets:
            new = append(lib, b, out, pos)
            assert new == pos + len(b)
            bounds.append((pos, new))
            pos = new
        assert pos == start + total
        for b, (lo, hi) in zip(buckets, bounds):
            assert sorted(out[lo:hi].tolist()) == sorted(b), "a bucket's groups must stay in the bucket, each exactly once"


def test_a_position_gets_its_class_whenever_the_bucket_has_one_left(lib):
    rng = random.Random(7)
    matched = total = 0
    for trial in range(300):
        n = rng.randrange(2, 40)
        pic = rng.randrange(0, 24)
        b = [entry(pic if trial % 2 else rng.randrange(0, 24), rng.randrange(0, 68), rng.randrange(0, 15)) for _ in range(n)]
        out = np.zeros(n + 16, dtype=np.uint32)
        pos0 = rng.randrange(0, 9)
        append(lib, b, out, pos0)
        left = {c: sum(1 for e in b if lib.co_class(e) == c) for c in range(4)}
        for i in range(n):
            p, c = pos0 + i, lib.co_class(int(out[pos0 + i]))
            if left[p & 3] > 0:
                assert c == (p & 3), (trial, i)
                matched += 1
            else:
                assert left[c] == max(left.values())        # nothing of the position's class left: the class the bucket holds most of
            left[c] -= 1
            total += 1
    assert matched > 0.7 * total


def test_a_balanced_bucket_is_dealt_exactly(lib):
    # one picture, fifteen columns of one key: four groups of every class except the last
    b = [entry(3, 20 - c, c) for c in range(15)]
    out = np.zeros(20, dtype=np.uint32)
    append(lib, b, out, 0)
    classes = [lib.co_class(int(e)) for e in out[:15]]
    assert classes[:12] == [0, 1, 2, 3] * 3
