"""Regenerates tests/golden/hevc_*.h265 and golden_hevc.json.

Same procedure as make_golden.py: streams come from the build's own seeded generator (tools/hevcgen.c); each expected MD5 is the
tight-I420 display-order output of the CPU oracle (oracle/orc_hevc_*.c) and is only written when the generator's independently
written reconstruction loop produced byte-identical frames.  The syntax digest is the oracle's FNV-1a over every coding unit's
syntax (the product's host parser must reproduce it).  Nothing here is third-party: no HEVC stream or decoder exists in this image,
so these vectors pin the oracle against regressions, not against the standard ("parity unpinned", oracle/orc_hevc.h).
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
    "hevc_intra_96x80": dict(width=96, height=80, frames=2, intra_period=1, seed=31),
    "hevc_ip_real_96x80": dict(width=96, height=80, frames=6, seed=32, num_ref=2),
    "hevc_ipb_fuzz_crop_90x70": dict(width=90, height=70, frames=7, gop=2, num_ref=2, mode=1, seed=33, ctb_log2=5, sdh=1, dqp=2, cb_qp_off=-3, cr_qp_off=4, deblock=2),
    "hevc_gop8_wpp_128x96": dict(width=128, height=96, frames=9, gop=8, num_ref=2, seed=34, ctb_log2=4, wpp=1, sdh=1),
    "hevc_tools_fuzz_96x80": dict(width=96, height=80, frames=5, mode=1, seed=35, ctb_log2=4, tskip=1, pcm=1, bypass=1, scaling=2, wp=1, rplm=1, num_ref=3, slice_ctus=7, dep_slices=1, cip=1),
    "hevc_tiles_lt_128x96": dict(width=128, height=96, frames=8, seed=36, ctb_log2=4, tile_cols=2, tile_rows=2, lt_ref=1, num_ref=2, scaling=3, merge_cand=3, par_mrg=4),
}


def main():
    o = streams.OracleHevc()
    meta = {}
    for name, kw in CASES.items():
        with tempfile.NamedTemporaryFile(suffix=".yuv") as tf:
            data = streams.generate_hevc(recon_path=tf.name, **kw)
            recon = open(tf.name, "rb").read()
        out, n, w, h = o.decode(data, 1)
        assert out == recon, f"{name}: oracle and generator reconstruction differ"
        nv12, _, _, _ = o.decode(data, 0)
        dig, ncu = o.syntax_digest(data)
        open(os.path.join(HERE, name + ".h265"), "wb").write(data)
        fs = w * h * 3 // 2
        meta[name] = {"params": kw, "frames": n, "width": w, "height": h, "bytes": len(data), "md5_i420": hashlib.md5(out).hexdigest(), "md5_nv12": hashlib.md5(nv12).hexdigest(),
                      "md5_frames_i420": [hashlib.md5(out[i * fs:(i + 1) * fs]).hexdigest() for i in range(n)], "syntax_digest": "%016x" % dig, "coding_units": ncu, "tools": o.tools(data)}
        print(name, len(data), "bytes", n, "frames", meta[name]["md5_i420"])
    json.dump(meta, open(os.path.join(HERE, "golden_hevc.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
