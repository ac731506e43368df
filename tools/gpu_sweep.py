#!/usr/bin/env python3
"""Developer tool: randomised differential test on a GPU box -- N random generator configurations per codec, decoded through the C ABI
on cuda:0 and compared bit-exactly with the CPU oracle (which itself must equal the generator's reconstruction).
    python tools/gpu_sweep.py [n] [seed] [big]      # big: picture sizes up to 1280x720 (several CTB rows / deblocking bands / XCD bands)"""
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jmcodec_amd                      # noqa: E402
from tools import streams         # noqa: E402


def hevc_params(r):
    ctb = r.choice([4, 5, 6])
    a = dict(width=r.choice([64, 96, 120, 176, 200, 130, 320]), height=r.choice([64, 80, 96, 144, 66, 240]), frames=r.choice([2, 3, 5, 9, 17]),
        qp=r.choice([18, 26, 32, 38, 44]),
             seed=r.randrange(1 << 20), gop=r.choice([0, 0, 1, 2, 3, 8]), num_ref=r.choice([1, 2, 3, 4]), ctb_log2=ctb, mode=r.choice([0, 1, 1]),
             amp=r.randint(0, 1), sao=r.randint(0, 1),
             deblock=r.choice([0, 1, 1, 2]), tskip=r.randint(0, 1), sdh=r.randint(0, 1), dqp=r.choice([0, 0, 1, 2, 3, 4]), pcm=r.choice([0, 0, 1, 2]),
             bypass=r.choice([0, 0, 1]),
             cip=r.choice([0, 0, 1]), tmvp=r.randint(0, 1), wp=r.choice([0, 0, 1]), rplm=r.choice([0, 0, 1]), scaling=r.choice([0, 0, 1, 2, 3]),
             wpp=r.choice([0, 0, 1]),
             min_cb_log2=r.choice([3, 3, min(4, ctb)]), max_tb_log2=r.choice([5, 5, 4, 3]), depth_inter=r.randint(0, 3), depth_intra=r.randint(0, 3),
             strong_intra=r.randint(0, 1),
             merge_cand=r.randint(1, 5), cabac_init=r.choice([0, 1, 2]), par_mrg=r.choice([2, 2, 3, 4, 5]), intra_period=r.choice([4, 8, 32]),
             cb_qp_off=r.choice([0, 0, -3, 5]),
             cr_qp_off=r.choice([0, 0, 4, -6]), rps_sps=r.randint(0, 1), open_gop=r.choice([0, 0, 1]))
    if r.random() < 0.25:
        a.update(tile_cols=r.randint(1, 3), tile_rows=r.randint(1, 3))
    if r.random() < 0.3:
        a.update(slice_ctus=r.randint(1, 9), dep_slices=r.randint(0, 1))
    if a["gop"] == 0 and r.random() < 0.3:
        a["lt_ref"] = 1
    return a


def h264_params(r):
    cab = r.randint(0, 1)
    b = r.choice([0, 0, 1, 2, 3]) if cab or r.random() < 0.5 else 0
    a = dict(width=r.choice([64, 96, 90, 176, 200, 320]), height=r.choice([48, 80, 70, 144, 240, 272, 400, 520]), frames=r.choice([2, 4, 7, 12]),
        # > 256 rows: several deblocking bands
             qp=r.choice([18, 24, 28, 36, 44]), gop=r.choice([2, 4, 6, 30]),
             seed=r.randrange(1 << 20), mode=r.choice([0, 1, 1]), deblock=r.choice([0, 1, 1, 2]), num_ref=r.randint(1, 4), slices=r.randint(1, 3), cabac=cab,
             cabac_idc=r.randint(0, 2),
             t8x8=r.randint(0, 1), bframes=b, direct_temporal=r.randint(0, 1), wp=r.choice([0, 0, 1, 2]), dinf8=r.randint(0, 1), scaling=r.choice([0, 0, 1,
             2]), rplm=r.choice([0, 0, 1]),
             cip=r.choice([0, 0, 1]), chroma_qp_off=r.choice([0, 0, -4, 6]), alpha_off=r.choice([0, 0, 3, -3]), beta_off=r.choice([0, 0, -2, 2]),
             poc_type=r.choice([0, 2]),
             nc_corner=r.choice([0, 0, 0, 1]), no_intra=r.choice([0, 1, 1]), search=r.choice([4, 4, 16,
             48]))      # no_intra: pictures that can run in chain launches
    if not b and r.random() < 0.3:
        a["mmco"] = 1
    if not b and r.random() < 0.3:
        # round 4: memory management operation 5, pic_order_cnt_type 1 (with a bottom-field offset and non-reference pictures in
        # the cycle) and long-term references TOGETHER, in frame streams too -- the combinations the three shared deviations of round 3 hid in
        a.update(mmco=2, poc_type=r.choice([0, 1, 1, 2]), poc_bottom=r.randint(0, 1), nonref_period=r.choice([0, 2, 3]), num_ref=r.randint(2, 4),
                 frames=r.choice([14, 20, 30]), gop=r.choice([10, 30]))
    if (cab or b) and r.random() < 0.25 and ((a["height"] + 15) // 16) % 2 == 0 and (((a["height"] + 15) // 16) * 16 - a["height"]) % 4 == 0:
        a.update(fmo0=1, dinf8=1)                                                                           # interlace-capable stream, frame pictures only
    if b and r.random() < 0.3 and ((a["height"] + 15) // 16) % 2 == 0 and (((a["height"] + 15) // 16) * 16 - a["height"]) % 4 == 0:
        a.update(paff=r.choice([1, 2]), dinf8=1)                                                                                    # B field pictures
        if a["cabac"]:
            a["t8x8"] = 0
    if not b and r.random() < 0.3 and ((a["height"] + 15) // 16) % 2 == 0 and (((a["height"] + 15) // 16) * 16 - a["height"]) % 4 == 0:
        a.update(paff=r.choice([1, 2]), poc_type=r.choice([0, 1, 2]), poc_bottom=r.randint(0, 1), nonref_period=r.choice([0, 0, 3]))      # field pictures
        if a.get("mmco"):
            a["mmco"] = r.choice([1, 2])
    if r.random() < 0.15:
        a["redundant"] = 1                                                                                            # slices of redundant coded pictures
    if not b and r.random() < 0.25:
        a["gaps"] = 1                                                                                                 # frame_num values nobody sends (8.2.5.2)
    if r.random() < 0.4:
        a["frames"] = r.choice([9, 14, 20])                                                                        # long enough for deep chains
    return a


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    base = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    oh, o4 = streams.OracleHevc(), streams.Oracle()
    bad = 0
    for codec, name in ((1, "hevc"), (0, "h264")):
        for i in range(n):
            r = random.Random(base * 100003 + i * 7 + codec)
            kw = hevc_params(r) if codec else h264_params(r)
            if os.environ.get("SWEEP_VERBOSE"):
                print(name, i, flush=True)
            if big:
                kw.update(width=r.choice([416, 640, 854, 1280, 720]), height=r.choice([240, 360, 480, 720, 576]), frames=r.choice([2, 3, 5]))
            with tempfile.NamedTemporaryFile(suffix=".yuv") as tf:
                data = (streams.generate_hevc if codec else streams.generate)(recon_path=tf.name, **kw)
                recon = open(tf.name, "rb").read()
            try:
                want = (oh if codec else o4).decode(data, 1)[0]
            except RuntimeError as e:
                print("ORACLE FAIL", name, kw, e); bad += 1; continue
            if want != recon:
                print("ORACLE != GENERATOR", name, kw); bad += 1; continue
            # NAL by NAL (pictures trickle into the engine), and the whole stream in one call with random chain knobs (every picture pending at once:
            # the engine forms chain launches of consecutive pictures, chain.hip)
            for whole in (False, True):
                with jmcodec_amd.JmAmdDec(codec, 1, options={"device": 0}) as d:
                    if whole:
                        L = jmcodec_amd.lib()
                        L.jm_amddec_set_option(d.h, b"chain_depth", r.choice([2, 3, 8, 16])); L.jm_amddec_set_option(d.h, b"chain_lag", r.choice([20, 24, 40]))
                    got = b"".join(d.decode_stream(None, chunks=[data]) if whole else d.decode_stream(data)); err = d.stat("errors")
                if got != want or err:
                    print("GPU MISMATCH", name, "whole" if whole else "nal", kw, "errors", err); bad += 1
        print(name, n, "configurations done, failures so far:", bad, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
