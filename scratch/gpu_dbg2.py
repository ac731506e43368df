import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jmcodec_amd
from tools import streams
cfgs = [
 {'width': 854, 'height': 480, 'frames': 2, 'qp': 18, 'seed': 764948, 'gop': 2, 'num_ref': 1, 'ctb_log2': 5, 'mode': 1, 'amp': 1, 'sao': 1, 'deblock': 0, 'tskip': 1, 'sdh': 1, 'dqp': 0, 'pcm': 0, 'bypass': 0, 'cip': 0, 'tmvp': 1, 'wp': 0, 'rplm': 0, 'scaling': 3, 'wpp': 0, 'min_cb_log2': 3, 'max_tb_log2': 5, 'depth_inter': 3, 'depth_intra': 1, 'strong_intra': 0, 'merge_cand': 1, 'cabac_init': 2, 'par_mrg': 3, 'intra_period': 8, 'cb_qp_off': 0, 'cr_qp_off': -6, 'rps_sps': 0, 'open_gop': 0, 'slice_ctus': 7, 'dep_slices': 0},
 {'width': 854, 'height': 480, 'frames': 3, 'qp': 44, 'seed': 715012, 'gop': 0, 'num_ref': 2, 'ctb_log2': 4, 'mode': 1, 'amp': 0, 'sao': 1, 'deblock': 2, 'tskip': 1, 'sdh': 0, 'dqp': 3, 'pcm': 0, 'bypass': 1, 'cip': 1, 'tmvp': 0, 'wp': 1, 'rplm': 0, 'scaling': 0, 'wpp': 0, 'min_cb_log2': 4, 'max_tb_log2': 5, 'depth_inter': 3, 'depth_intra': 1, 'strong_intra': 0, 'merge_cand': 1, 'cabac_init': 1, 'par_mrg': 5, 'intra_period': 4, 'cb_qp_off': 5, 'cr_qp_off': 4, 'rps_sps': 0, 'open_gop': 1, 'slice_ctus': 7, 'dep_slices': 0},
]
for kw in cfgs:
    with tempfile.NamedTemporaryFile(suffix=".yuv") as tf:
        data = streams.generate_hevc(recon_path=tf.name, **kw); recon = open(tf.name, "rb").read()
    want = streams.OracleHevc().decode(data, 1)[0]
    with jmcodec_amd.JmAmdDec(1, 1, options={"device": 0}) as d:
        got = b"".join(d.decode_stream(data))
    print("product == oracle:", got == want, " product == generator:", got == recon, " oracle == generator:", want == recon)
