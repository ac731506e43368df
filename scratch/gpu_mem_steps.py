# where does the resident memory of a bench rank come from?  (developer script, run on the GPU box)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def rss(tag):
    f = {l.split(":")[0]: l.split(":")[1].strip() for l in open("/proc/self/status") if l.startswith(("VmRSS", "RssAnon", "RssFile"))}
    print(f"{tag:40s} {f}", flush=True)
rss("start")
import numpy
rss("numpy")
import torch
rss("import torch")
import torch.distributed as dist
rss("torch.distributed")
import jmcodec_amd
from jmcodec_amd import api
rss("import jmcodec_amd (library loaded)")
from tools import streams
data = streams.generate(**streams.config_c1(stream_id=0, frames=30))
rss("one 1080p stream generated")
d = api.JmAmdDec(0, 1); d.__enter__()
rss("first handle created + init (engine up)")
frames = d.decode_stream(data)
rss("30 frames decoded (1 handle)")
hs = []
for i in range(7):
    h = api.JmAmdDec(0, 1); h.__enter__(); h.decode_stream(data, keep=False); hs.append(h)
rss("8 handles, each decoded 30 frames")
if len(sys.argv) > 1:
    torch.cuda.init(); torch.zeros(1, device="cuda:0"); rss("torch.cuda initialised")
