import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import jmcodec_amd
from jmcodec_amd import streams, api
from test_hevc_oracle import HEVC_CASES
data = streams.generate_hevc(**HEVC_CASES["b_gop8"])
rec, packets = api.annexb_to_hvcc(data)
for opts in ({"device": 0}, {"parse_only": 1}):
    with jmcodec_amd.JmAmdDec(1, 1, options=opts, extra_data=rec) as d:
        frames = d.decode_stream(None, chunks=packets)
        print(opts, len(frames), d.stat("errors"), d.stat("pictures"), jmcodec_amd.lib().jm_amddec_last_error(d.h))
