"""Developer script (round 6): the 32-stream short-GOP scenario of tests/test_gpu_parity.py::test_32_streams_with_intra_pictures_ahead_of_their_turn as a stand-alone program
(bisecting a crash by environment switches)."""
import sys, os, threading, ctypes as C, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmcodec_amd import api
from tools import streams
kinds = [dict(width=640, height=368, frames=72, gop=12, seed=900 + k, num_ref=1 + k % 2) for k in range(4)]
datas = [streams.generate(**kinds[i % 4]) for i in range(32)]
lib = api.lib()
def run(i, d):
    h = d.h; gotf = C.c_int(0); n = 0
    dev, ln = C.c_void_p(0), C.c_int(0)
    for nal in api.split_nalus(datas[i]):
        lib.jm_amddec_decode_frame(C.cast(C.c_char_p(nal), C.c_void_p), len(nal), C.byref(gotf), h)
        if gotf.value == 1 and lib.jm_amddec_output_frame_device(C.byref(dev), C.byref(ln), h) > 0: n += 1
    while not lib.jm_amddec_is_exit(h):
        if lib.jm_amddec_decode_frame(None, 0, C.byref(gotf), h) != 0: break
        if gotf.value == 1: n += 1
    res[i] = (n, d.stat("errors"))
res = [None] * 32
decs = [api.JmAmdDec(0, 1, options={"device_output": 1}) for _ in range(32)]
ts = [threading.Thread(target=run, args=(i, decs[i])) for i in range(32)]
[t.start() for t in ts]; [t.join() for t in ts]
print("early", decs[0].stat("eng_early_intra"), "regrown", sum(d.stat("job_regrown") for d in decs), res[:4])
for d in decs: d.close()
print("ok")
