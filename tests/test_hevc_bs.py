"""HEVC boundary strengths on the CPU: the map painting and the per-edge derivation that k_hevc_bs_raster / k_hevc_bs run on the device
(jmcodec_amd/csrc/hevc_bs.h, host build) against a LITERAL restatement of ITU-T H.265 8.7.2.3 (which edges are transform / prediction block edges) and
8.7.2.4 (bS 2 / 1 / 0) that works from the coding blocks themselves -- a random coding quadtree per CTB with intra and inter coding units, every partition
mode, random transform trees and coded-block flags -- not from the product's maps or job lists."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDB_DISABLED, HDB_CONCEALED, HDB_NO_LEFT, HDB_NO_TOP = 1, 2, 4, 8


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libhevc_bs_check.so")
    src = os.path.join(ROOT, "tests", "native", "hevc_bs_check.cpp")
    hdrs = [os.path.join(ROOT, "jmcodec_amd", "csrc", h) for h in ("hevc_bs.h", "hevc_jobs.h", "jobs.h", "mc_packed.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-unused-function", "-o", so, src])
    l = ctypes.CDLL(so)
    l.hbs_run.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
    return l


class Picture:
    """A random partitioning of a w x h picture into coding units, prediction blocks and transform blocks (sample-accurate maps of who owns what)."""

    def __init__(self, rng, w, h, ctb_log2):
        self.w, self.h, self.lg = w, h, ctb_log2
        self.cu = -np.ones((h, w), dtype=np.int64)          # coding unit id per sample
        self.pb = -np.ones((h, w), dtype=np.int64)          # prediction block id (inter), -1 in intra units
        self.tb = -np.ones((h, w), dtype=np.int64)          # transform block id
        self.intra, self.motion, self.cbf, self.pbs, self.tbs = [], [], [], [], []
        self.rng = rng
        # a small pool of motions, so that equal and nearly equal ones meet: (slot0, slot1, mv0, mv1)
        self.pool = []
        for _ in range(6):
            kind = rng.integers(0, 3)
            s0 = int(rng.integers(0, 3)) if kind != 1 else -1
            s1 = int(rng.integers(0, 3)) if kind != 0 else -1
            self.pool.append((s0, s1, tuple(int(v) for v in rng.integers(-6, 7, 2)), tuple(int(v) for v in rng.integers(-6, 7, 2))))
        cs = 1 << ctb_log2
        for y in range(0, h, cs):
            for x in range(0, w, cs):
                self.quad(x, y, cs)

    def quad(self, x, y, n):
        if x >= self.w or y >= self.h:
            return
        if n > 8 and (x + n > self.w or y + n > self.h or self.rng.integers(0, 3) > 0 or n > 32):
            for dy in (0, n // 2):
                for dx in (0, n // 2):
                    self.quad(x + dx, y + dy, n // 2)
            return
        self.coding_unit(x, y, n)

    def coding_unit(self, x, y, n):
        rng = self.rng
        cid = len(self.intra)
        intra = bool(rng.integers(0, 4) == 0)
        self.intra.append(intra)
        self.cu[y:y + n, x:x + n] = cid
        parts = [(0, 0, n, n)]
        if not intra:
            mode = rng.integers(0, 8 if n >= 16 else 3)        # 2Nx2N, 2NxN, Nx2N, and with n >= 16 the four asymmetric ones (+ 2Nx2N again)
            q = n // 4
            parts = {0: [(0, 0, n, n)], 1: [(0, 0, n, n // 2), (0, n // 2, n, n // 2)], 2: [(0, 0, n // 2, n), (n // 2, 0, n // 2, n)],
                     3: [(0, 0, n, q), (0, q, n, n - q)], 4: [(0, 0, n, n - q), (0, n - q, n, q)], 5: [(0, 0, q, n), (q, 0, n - q, n)],
                     6: [(0, 0, n - q, n), (n - q, 0, q, n)], 7: [(0, 0, n, n)]}[int(mode)]
            for (px, py, pw, ph) in parts:
                pid = len(self.pbs)
                self.pbs.append((x + px, y + py, pw, ph, self.pool[int(rng.integers(0, len(self.pool)))]))
                self.pb[y + py:y + py + ph, x + px:x + px + pw] = pid
        self.transform_tree(x, y, min(n, 32), n, intra, depth=0)

    def transform_tree(self, x, y, t, n, intra, depth):
        if n > t:                                              # a 64x64 unit: the transform tree starts at 32x32
            for dy in range(0, n, t):
                for dx in range(0, n, t):
                    self.transform_tree(x + dx, y + dy, t, t, intra, depth)
            return
        if t > 4 and depth < 2 and self.rng.integers(0, 3) == 0:
            for dy in (0, t // 2):
                for dx in (0, t // 2):
                    self.transform_tree(x + dx, y + dy, t // 2, t // 2, intra, depth + 1)
            return
        tid = len(self.tbs)
        self.tbs.append((x, y, t, intra, bool(self.rng.integers(0, 2))))
        self.tb[y:y + t, x:x + t] = tid

    # ---- what the host puts into the job lists (hevc_slice.cpp): prediction blocks cut into pieces of at most 16x16, luma intra blocks, coded luma blocks ----
    def job_lists(self):
        pus, itbs, tbs = [], [], []
        for (x, y, w, h, (s0, s1, m0, m1)) in self.pbs:
            for yy in range(y, y + h, 16):
                for xx in range(x, x + w, 16):
                    pus.append([xx, yy, min(16, x + w - xx), min(16, y + h - yy), s0, s1, m0[0], m0[1], m1[0], m1[1]])
        for (x, y, t, intra, cbf) in self.tbs:
            lg = t.bit_length() - 1
            if intra:
                itbs.append([x, y, lg])
            elif cbf:
                tbs.append([x, y, lg])
        return pus, itbs, tbs

    # ---- 8.7.2.3 + 8.7.2.4, literally, for the edge segment whose first q sample is (x, y) ----
    def literal(self, dirn, x, y, db_flags, qp8):
        xp, yp = (x, y - 1) if dirn else (x - 1, y)
        if (y == 0) if dirn else (x == 0):
            return 0                                            # picture boundary: filterEdgeFlag 0
        lg, cw = self.lg, (self.w + (1 << self.lg) - 1) >> self.lg
        fq, fp = db_flags[(y >> lg) * cw + (x >> lg)], db_flags[(yp >> lg) * cw + (xp >> lg)]
        if fq & (HDB_DISABLED | HDB_CONCEALED):
            return 0
        if ((y if dirn else x) & ((1 << lg) - 1)) == 0 and (fq & (HDB_NO_TOP if dirn else HDB_NO_LEFT)):
            return 0
        tq, tp = self.tb[y, x], self.tb[yp, xp]
        pq, pp_ = self.pb[y, x], self.pb[yp, xp]
        iq, ip = self.intra[self.cu[y, x]], self.intra[self.cu[yp, xp]] or bool(fp & HDB_CONCEALED)
        transform_edge = tq != tp
        prediction_edge = self.cu[y, x] != self.cu[yp, xp] or pq != pp_
        if not transform_edge and not prediction_edge:
            return 0                                            # not an edge of the edge set
        if iq or ip:
            bs = 2
        elif transform_edge and ((not self.tbs[tq][3] and self.tbs[tq][4]) or (not self.tbs[tp][3] and self.tbs[tp][4])):
            bs = 1                                              # a transform block edge with a non-zero coefficient on either side
        else:
            a, b = self.pbs[pq][4], self.pbs[pp_][4]
            bs = 1 if self.motion_differs(a, b) else 0
        if bs:
            w8 = self.w >> 3
            if qp8[(yp >> 3) * w8 + (xp >> 3)] & 128:
                bs |= 4
            if qp8[(y >> 3) * w8 + (x >> 3)] & 128:
                bs |= 8
        return bs

    @staticmethod
    def motion_differs(a, b):
        """8.7.2.4, the motion conditions: different reference pictures or numbers of vectors; one vector each: a component differs by >= 4; two each (to
        the same two pictures): both pairings tested as the clause lists them."""
        refs_a = sorted(s for s in a[:2] if s >= 0)
        refs_b = sorted(s for s in b[:2] if s >= 0)
        if refs_a != refs_b:
            return True
        far = lambda u, v: abs(u[0] - v[0]) >= 4 or abs(u[1] - v[1]) >= 4
        mv = lambda m: [m[2 + l] for l in range(2) if m[l] >= 0]
        if len(refs_a) == 1:
            return far(mv(a)[0], mv(b)[0])
        if refs_a[0] != refs_a[1]:                              # two different reference pictures: compare the vectors that point to the same picture
            bv = {b[0]: b[2], b[1]: b[3]}
            return far(a[2], bv[a[0]]) or far(a[3], bv[a[1]])
        # both vectors refer to the same picture: differs only when BOTH pairings differ
        return (far(a[2], b[2]) or far(a[3], b[3])) and (far(a[2], b[3]) or far(a[3], b[2]))


@pytest.mark.parametrize("seed,w,h,lg", [(1, 128, 128, 6), (2, 192, 128, 6), (3, 96, 64, 5), (4, 128, 96, 4), (5, 256, 128, 6), (6, 64, 64, 6), (7, 160, 96, 5)])
def test_strengths_against_the_literal_clause(lib, seed, w, h, lg):
    rng = np.random.default_rng(seed)
    for trial in range(6):
        pic = Picture(rng, w, h, lg)
        cw, ch = (w + (1 << lg) - 1) >> lg, (h + (1 << lg) - 1) >> lg
        db = np.zeros(cw * ch, dtype=np.uint8)
        if trial >= 2:                                         # slice / tile boundaries that must not be filtered across, a slice with the filter off, a concealed CTB
            db[:] = rng.choice([0, 0, 0, HDB_NO_LEFT, HDB_NO_TOP, HDB_NO_LEFT | HDB_NO_TOP, HDB_DISABLED, HDB_CONCEALED], size=cw * ch)
        qp8 = (rng.integers(20, 40, size=(w >> 3) * (h >> 3)) | (rng.integers(0, 8, size=(w >> 3) * (h >> 3)) == 0) * 128).astype(np.uint8)
        pus, itbs, tbs = pic.job_lists()
        P = np.array(pus, dtype=np.int32).reshape(-1, 10); I = np.array(itbs, dtype=np.int32).reshape(-1, 3); T = np.array(tbs, dtype=np.int32).reshape(-1, 3)
        w8, w4, h4, h8 = w >> 3, w >> 2, h >> 2, h >> 3
        bs_v = np.zeros(w8 * h4, dtype=np.uint8); bs_h = np.zeros(w4 * h8, dtype=np.uint8)
        lib.hbs_run(w, h, lg, P.ctypes.data, len(P), I.ctypes.data, len(I), T.ctypes.data, len(T), db.ctypes.data, qp8.ctypes.data, bs_v.ctypes.data, bs_h.ctypes.data)
        # a concealed CTB's samples count as intra towards its neighbours; the lists hold nothing for it: give the literal side the same view
        want_v = np.array([pic.literal(0, (i % w8) * 8, (i // w8) * 4, db, qp8) for i in range(w8 * h4)], dtype=np.uint8)
        want_h = np.array([pic.literal(1, (i % w4) * 4, (i // w4) * 8, db, qp8) for i in range(w4 * h8)], dtype=np.uint8)
        bad = np.flatnonzero(bs_v != want_v)
        assert not len(bad), (seed, trial, "vertical", int(bad[0]), int(bs_v[bad[0]]), int(want_v[bad[0]]))
        bad = np.flatnonzero(bs_h != want_h)
        assert not len(bad), (seed, trial, "horizontal", int(bad[0]), int(bs_h[bad[0]]), int(want_h[bad[0]]))
        if trial < 2 and w * h >= 96 * 64:
            assert (want_v & 3).max() == 2 and ((want_v & 3) == 1).any() and ((want_v & 3) == 0).any()       # every strength occurs
