# resident memory of a process after plain HIP runtime calls (no decoder): what the runtime itself costs on this box
import ctypes as C
def rss(tag):
    f = {l.split(":")[0]: l.split(":")[1].strip() for l in open("/proc/self/status") if l.startswith(("VmRSS", "RssAnon"))}
    print(f"{tag:40s} {f}", flush=True)
rss("start")
hip = C.CDLL("libamdhip64.so")
rss("libamdhip64 loaded")
print("hipInit", hip.hipInit(0)); rss("hipInit")
n = C.c_int(0); hip.hipGetDeviceCount(C.byref(n)); rss("hipGetDeviceCount %d" % n.value)
print("hipSetDevice", hip.hipSetDevice(0)); rss("hipSetDevice(0)")
p = C.c_void_p(); print("hipMalloc", hip.hipMalloc(C.byref(p), 1 << 20)); rss("hipMalloc 1 MB")
s = C.c_void_p(); print("hipStreamCreate", hip.hipStreamCreate(C.byref(s))); rss("hipStreamCreate")
q = C.c_void_p(); print("hipHostMalloc", hip.hipHostMalloc(C.byref(q), 64 << 20, 0)); rss("hipHostMalloc 64 MB")
import ctypes
ctypes.memset(q, 1, 64 << 20); rss("touched")
for i in range(10):
    s2 = C.c_void_p(); hip.hipStreamCreate(C.byref(s2)); rss("stream %d" % (i + 2))
for i in range(3):
    s3 = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s3), 1); rss("non-blocking stream %d" % i)
