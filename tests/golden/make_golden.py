"""Regenerates tests/golden/*.h264 and golden.json.

The reference holds no fixtures of any kind, so these vectors come from the
build's own seeded generator (tools/h264gen.c).  Each expected MD5 is the tight-I420 display-order
output of the CPU oracle, and is only written when the generator's independent reconstruction loop
produced byte-identical frames (two separately written code paths agree).  pcm_* streams are true
known-answer vectors: decoded samples equal the I_PCM payload bytes in the stream.
"""
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tools import streams  # noqa: E402

CASES = {
    "pcm_64x48": dict(width=64, height=48, frames=2, pcm_only=1, gop=2, deblock=0, seed=11),
    "pcm_db_48x32": dict(width=48, height=32, frames=2, pcm_only=1, gop=1, deblock=1, seed=12),
    "intra_64x48": dict(width=64, height=48, frames=2, gop=1, deblock=1, seed=13),
    "ip_real_96x80": dict(width=96, height=80, frames=6, gop=6, deblock=1, seed=14),
    "ip_fuzz_96x80": dict(width=96, height=80, frames=8, gop=4, mode=1, num_ref=3, slices=2, seed=15),
    "ip_fuzz_crop_90x70": dict(width=90, height=70, frames=6, gop=6, mode=1, num_ref=2, seed=16, poc_type=0, nonref_period=3, deblock=2, slices=3),
    "ip_fuzz_cip_80x64": dict(width=80, height=64, frames=6, gop=6, mode=1, num_ref=4, seed=17, cip=1, chroma_qp_off=-4, alpha_off=3, beta_off=-3),
    "main_cabac_fuzz_96x80": dict(width=96, height=80, frames=8, gop=4, mode=1, num_ref=3, slices=2, seed=18, cabac=1, cabac_idc=1),
    "high_cabac_fuzz_96x80": dict(width=96, height=80, frames=8, gop=4, mode=1, num_ref=2, seed=19, cabac=1, cabac_idc=2, t8x8=1),
    "high_cavlc_real_96x80": dict(width=96, height=80, frames=6, gop=6, seed=20, t8x8=1),
}


def main():
    o = streams.Oracle()
    meta = {}
    for name, kw in CASES.items():
        with tempfile.NamedTemporaryFile(suffix=".yuv") as tf:
            data = streams.generate(recon_path=tf.name, **kw)
            recon = open(tf.name, "rb").read()
        out, n, w, h = o.decode(data, 1)
        assert out == recon, f"{name}: oracle and generator reconstruction differ"
        nv12, n2, _, _ = o.decode(data, 0)
        open(os.path.join(HERE, name + ".h264"), "wb").write(data)
        fs = w * h * 3 // 2
        meta[name] = {
            "params": kw, "frames": n, "width": w, "height": h, "bytes": len(data),
            "md5_i420": hashlib.md5(out).hexdigest(), "md5_nv12": hashlib.md5(nv12).hexdigest(),
            "md5_frames_i420": [hashlib.md5(out[i * fs:(i + 1) * fs]).hexdigest() for i in range(n)],
        }
        print(name, len(data), "bytes", n, "frames", meta[name]["md5_i420"])
    json.dump(meta, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
