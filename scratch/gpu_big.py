import sys, random, subprocess, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import gpu_sweep
if len(sys.argv) > 3:
    import jmcodec_amd
    from tools import streams
    codec, base, i = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    r = random.Random(base * 100003 + i * 7 + codec)
    kw = gpu_sweep.hevc_params(r) if codec else gpu_sweep.h264_params(r)
    kw.update(width=r.choice([416, 640, 854, 1280, 720]), height=r.choice([240, 360, 480, 720, 576]), frames=r.choice([2, 3, 5]))
    print(kw, flush=True)
    data = (streams.generate_hevc if codec else streams.generate)(**kw)
    want = (streams.OracleHevc() if codec else streams.Oracle()).decode(data, 1)[0]
    with jmcodec_amd.JmAmdDec(codec, 1, options={"device": 0}) as d:
        got = b"".join(d.decode_stream(data)); err = d.stat("errors")
    print("RESULT", "ok" if got == want and not err else "MISMATCH", err, flush=True)
else:
    base = int(sys.argv[1]); n = int(sys.argv[2])
    for codec in (1, 0):
        for i in range(n):
            p = subprocess.run([sys.executable, __file__, str(codec), str(base), str(i)], capture_output=True, text=True, timeout=600)
            tail = (p.stdout.strip().splitlines() or [""])[-1]
            if p.returncode != 0 or "RESULT ok" not in tail:
                print("FAIL codec", codec, "i", i, "rc", p.returncode, p.stdout[-600:], p.stderr[-800:], flush=True)
            else:
                print("ok", codec, i, flush=True)
