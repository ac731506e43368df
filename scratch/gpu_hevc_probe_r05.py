"""round 5 (developer): phase timing of k_hevc_intra -- an all-intra 4K C3-style stream through the instrumented library (jmcodec_amd/lib_dbg_probe,
built from a temporary copy of hevc_kernels.hip with wall-clock probes; not part of the product)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from tools import streams
import jmcodec_amd
cfg = streams.config_c3(frames=4, width=3840, height=2160, stream_id=0); cfg.update(intra_period=1, gop=1)
data = streams.generate_hevc(**cfg)
print("stream bytes", len(data), flush=True)
with jmcodec_amd.JmAmdDec(1, 1, options={"device": 0}) as d:
    t0 = time.time(); frames = d.decode_stream(data, keep=False); print("frames", frames, "in", round(time.time() - t0, 3), "s", flush=True)
