#!/usr/bin/env python3
"""Developer tool, no GPU: randomised three-way check of the syntax layer and the reconstruction on the CPU -- N random generator configurations per codec
(the same distributions as tools/gpu_sweep.py):
  * the generator's own reconstruction must equal the oracle's decoded output byte for byte (two independently structured reconstructions);
  * the product's host parser (parse-only handle) must deliver the same number of frames without errors, and its per-element syntax digest must equal the
    oracle's (a third implementation of the syntax layer: final motion vectors, modes, QPs, coefficient levels).
    python tools/cpu_sweep.py [n] [seed] [workers] [big]      # big: picture sizes up to 1280x720, 2 - 5 frames"""
import os
import random
import sys
import tempfile
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(job):
    codec, base, i, big = job
    import jmcodec_amd
    from jmcodec_amd import api
    from tools import streams
    from tools.gpu_sweep import hevc_params, h264_params
    r = random.Random(base * 100003 + i * 7 + codec)
    kw = hevc_params(r) if codec else h264_params(r)
    name = "hevc" if codec else "h264"
    if big:
        kw.update(width=r.choice([416, 640, 854, 1280, 720]), height=r.choice([240, 360, 480, 720, 576]), frames=r.choice([2, 3, 5]))
    with tempfile.NamedTemporaryFile(suffix=".yuv") as tf:
        data = (streams.generate_hevc if codec else streams.generate)(recon_path=tf.name, **kw)
        recon = open(tf.name, "rb").read()
    orc = streams.OracleHevc() if codec else streams.Oracle()
    try:
        want, n = orc.decode(data, 1)[:2]
    except RuntimeError as e:
        return f"ORACLE FAIL {name} {kw} {e}"
    if want != recon:
        return f"ORACLE != GENERATOR {name} {kw}"
    od = orc.syntax_digest(data)
    with api.JmAmdDec(codec, 1, options={"parse_only": 1, "digest": 1}) as d:
        frames = d.decode_stream(data, keep=False)
        pd = (d.stat("syntax_digest") & (2 ** 64 - 1), d.stat("digest_mbs")); errors = d.stat("errors")
    if errors or frames != n:
        return f"HOST PARSER {name} {kw}: {frames} frames of {n}, errors {errors}"
    if tuple(pd) != tuple(od):
        return f"SYNTAX DIGEST {name} {kw}: product {pd} oracle {od}"
    return None


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    base = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    big = len(sys.argv) > 4 and sys.argv[4] == "big"
    bad = 0
    with Pool(workers, maxtasksperchild=2000) as pool:       # (bounds whatever a native library of the tool chain might keep per call)
        for codec, name in ((1, "hevc"), (0, "h264")):
            for res in pool.imap_unordered(one, [(codec, base, i, big) for i in range(n)], chunksize=4):
                if res:
                    print(res, flush=True); bad += 1
            print(name, n, "configurations done, failures so far:", bad, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
