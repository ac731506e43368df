import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jmcodec_amd
from tools import streams
cfgs = [
 dict(width=320, height=520, frames=7, qp=18, gop=4, seed=153603, mode=1, deblock=0, num_ref=3, slices=1, cabac=1, cabac_idc=0, t8x8=1, bframes=2, direct_temporal=0, wp=1, dinf8=0, scaling=0, rplm=1, cip=1, chroma_qp_off=-4, alpha_off=3, beta_off=0, poc_type=0),
 dict(width=640, height=720, frames=2, qp=18, gop=4, seed=686082, mode=1, deblock=1, num_ref=2, slices=3, cabac=1, cabac_idc=2, t8x8=1, bframes=1, direct_temporal=1, wp=1, dinf8=1, scaling=0, rplm=0, cip=1, chroma_qp_off=-4, alpha_off=-3, beta_off=0, poc_type=0),
 dict(width=1280, height=720, frames=5, qp=18, gop=6, seed=801820, mode=1, deblock=0, num_ref=4, slices=1, cabac=1, cabac_idc=2, t8x8=0, bframes=1, direct_temporal=0, wp=0, dinf8=0, scaling=2, rplm=0, cip=1, chroma_qp_off=0, alpha_off=3, beta_off=0, poc_type=2),
]
for kw in cfgs:
    data = streams.generate(**kw)
    want = streams.Oracle().decode(data, 1)[0]
    with jmcodec_amd.JmAmdDec(0, 1, options={"device": 0}) as d:
        got = b"".join(d.decode_stream(data))
    w, h = kw["width"], kw["height"]; fs = w * h * 3 // 2
    a = np.frombuffer(want, np.uint8); b = np.frombuffer(got, np.uint8)
    print(kw["width"], kw["height"], "len", len(a), len(b))
    for f in range(len(a) // fs):
        fa, fb = a[f*fs:(f+1)*fs], b[f*fs:(f+1)*fs]
        if np.array_equal(fa, fb): print(" frame", f, "ok"); continue
        Y = (fa[:w*h] != fb[:w*h]).reshape(h, w); ys, xs = np.nonzero(Y)
        print(" frame", f, "luma diffs", Y.sum(), "bbox", (xs.min(), ys.min(), xs.max(), ys.max()) if Y.sum() else None, "first MB rows", sorted(set((ys // 16).tolist()))[:12])
        C = (fa[w*h:] != fb[w*h:]); print("   chroma diffs", C.sum())
        if Y.sum():
            y0, x0 = ys.min(), xs[ys == ys.min()].min(); print("   first diff at", x0, y0, "MB", x0 // 16, y0 // 16, "want", fa[y0*w+x0:y0*w+x0+8], "got", fb[y0*w+x0:y0*w+x0+8])
        pass
