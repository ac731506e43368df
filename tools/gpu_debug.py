"""Developer script: decode generated streams on the GPU and report the first mismatch vs the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jmcodec_amd
from tools import streams

o = streams.Oracle()
cases = [
    dict(width=64, height=48, frames=2, pcm_only=1, gop=2, deblock=0),
    dict(width=64, height=48, frames=1, gop=1, deblock=0),
    dict(width=64, height=48, frames=1, gop=1, deblock=1),
    dict(width=64, height=48, frames=3, gop=3, deblock=0),
    dict(width=64, height=48, frames=3, gop=3, deblock=1),
    dict(width=96, height=80, frames=6, gop=6, mode=1, deblock=0, seed=77),
    dict(width=96, height=80, frames=6, gop=6, mode=1, deblock=1, num_ref=3, slices=2, seed=77),
    dict(width=176, height=144, frames=8, gop=8, mode=1, num_ref=2, seed=79, cip=1, chroma_qp_off=-3, alpha_off=2, beta_off=-2),
    dict(width=320, height=240, frames=10, gop=10, seed=5),
    dict(width=1920, height=1080, frames=4, gop=4, seed=0x4A4D0100),
]
bad = 0
for kw in cases:
    data = streams.generate(**kw)
    want, n, w, h = o.decode(data, 1)
    with jmcodec_amd.JmAmdDec(0, 1) as d:
        frames = d.decode_stream(data)
        errs = d.stat("errors")
    got = b"".join(frames)
    ok = got == want and len(frames) == n
    print(("OK  " if ok else "FAIL"), kw, "frames", len(frames), "/", n, "errors", errs, flush=True)
    if not ok:
        bad += 1
        fs = w * h * 3 // 2
        for i in range(min(len(frames), n)):
            a = np.frombuffer(frames[i], np.uint8); b = np.frombuffer(want[i * fs:(i + 1) * fs], np.uint8)
            if not np.array_equal(a, b):
                idx = np.flatnonzero(a != b)
                k = int(idx[0])
                if k < w * h: where = f"Y x={k % w} y={k // w} (mb {k % w // 16},{k // w // 16})"
                else:
                    kk = k - w * h; pl = kk // (w * h // 4); kk %= (w * h // 4)
                    where = f"{'UV'[pl]} x={kk % (w // 2)} y={kk // (w // 2)} (mb {kk % (w // 2) // 8},{kk // (w // 2) // 8})"
                print(f"   frame {i}: {len(idx)} bytes differ, first at {where}: got {a[k]} want {b[k]}")
                ys = idx[idx < w * h]
                if len(ys):
                    mbs = sorted(set(((int(v) % w) // 16, (int(v) // w) // 16) for v in ys[:4000]))
                    print("   luma mbs (x,y):", mbs[:24])
                break
print("failures:", bad)
sys.exit(1 if bad else 0)
