#!/usr/bin/env python3
"""bench.py -- decoded frames/sec @1080p H.264 on N x MI355X (BASELINE.json metric), one process per GPU.

A "step" is one pass of the hot path over one batch: S independent 1080p Baseline I/P streams
(BASELINE config 1 / SURVEY 8d "C1", the per-GPU slice of config C4) of F frames each, decoded
concurrently by S jm_nvdec handles on this rank's GPU, fed NAL-by-NAL exactly like the reference
harness (test_nv_dec.cpp:184-250).  `value` = frames displayed by all ranks / wall time of the K timed
steps (barrier + device sync on both sides, max over ranks).

The boundary hands over HOST buffers (Annex-B bytes in, tight YUV out), so `value` is the end-to-end rate:
host entropy decode + H2D job lists + kernels + pack-out + D2H + the memcpy into the caller's buffer.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # the engine's streams need distinct HW queues (must precede HIP init, incl. torch's)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("JM_BENCH_STREAMS", "32")))
    ap.add_argument("--frames", type=int, default=60, help="frames per stream per step (multiple of the GOP, 30)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tools", default="baseline", choices=["baseline", "high", "high_b"],
                    help="diagnostic: coding tools of the synthetic stream (default = BASELINE config 1; high = CABAC + 8x8 transform; high_b = + I B B P)")
    ap.add_argument("--codec", default="h264", choices=["h264", "hevc"], help="diagnostic: hevc = SURVEY 8d config C3 (HEVC Main, 64x64 CTB, SAO + deblocking, random-access GOP 8) at --width x --height")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="diagnostic: do not record per-kernel HIP events")
    ap.add_argument("--parse-only", action="store_true", help="diagnostic: host stages only (no device work, frames carry no pixels)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    dist = None
    backend = os.environ.get("JM_BENCH_BACKEND", "nccl")      # "gloo" lets the multi-process path be exercised on a 1-GPU box
    n_dev = torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    os.environ["JM_AMD_DEC_DEVICE"] = str(local_rank % max(n_dev, 1))
    if world > 1 and "JM_AMD_DEC_THREADS" not in os.environ:      # share the host cores between the ranks of this node
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        cpus = os.cpu_count() or 8
        try:                                                       # a container may own far fewer CPUs than it sees (cgroup v2 quota)
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                cpus = min(cpus, int(int(q) * 1.75 / int(per) + 0.5))
        except Exception:
            pass
        os.environ["JM_AMD_DEC_THREADS"] = str(max(4, min(64, cpus // max(local_world, 1))))
    red_dev = "cuda" if (world > 1 and backend == "nccl") else "cpu"

    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "jmcodec_amd", "lib", "libjm_amd_dec.so")):
        ge.build()
    import jmcodec_amd
    from tools import streams
    L = jmcodec_amd.lib()
    if not jmcodec_amd.jm_nvdec_is_hw_support() and not args.parse_only:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")

    # ---- synthetic input: SURVEY 8d C1, seed = 0x4A4D0000 + 1*256 + stream_id (stream_id = rank) ----
    cfg = streams.config_c1(stream_id=rank, frames=args.frames, width=args.width, height=args.height)
    tools_desc = "Baseline, I/P-only (CAVLC"
    if args.tools != "baseline":
        cfg.update(cabac=1, t8x8=1)
        tools_desc = "High, I/P-only (CABAC, 8x8 transform"
    if args.tools == "high_b":
        cfg.update(bframes=2, num_ref=2, poc_type=0)
        tools_desc = "High, I B B P (CABAC, 8x8 transform"
    if args.codec == "hevc":
        cfg = streams.config_c3(frames=args.frames, width=args.width, height=args.height, stream_id=rank)
        tools_desc = "Main, random-access GOP 8 (CABAC, 64x64 CTB, SAO"
        data = streams.generate_hevc(**cfg)
    else:
        data = streams.generate(**cfg)
    nalus = jmcodec_amd.split_nalus(data)
    S, F, K, W = args.streams, args.frames, args.steps, args.warmup
    mb_w, mb_h = (args.width + 15) // 16, (args.height + 15) // 16
    frame_bytes = args.width * args.height * 3 // 2

    handles = []
    for _ in range(S):
        h = jmcodec_amd.jm_nvdec_create_handle()
        L.jm_amddec_set_option(h, b"profile", 0 if args.no_profile else 1)   # the engine records HIP events around each batched launch
        if args.parse_only:
            L.jm_amddec_set_option(h, b"parse_only", 1)
        if jmcodec_amd.jm_nvdec_init(1 if args.codec == "hevc" else 0, 1, None, 0, h) != 0:
            raise SystemExit("init failed: " + L.jm_amddec_last_error(h).decode())
        handles.append(h)

    counts = [0] * S

    def run_passes(i, passes):
        """test_nv_dec's hot loop: one NAL per jm_nvdec_decode_frame call, pull a frame whenever got_frame == 1."""
        h = handles[i]
        out = C.create_string_buffer(frame_bytes)
        got = C.c_int(0)
        n = C.c_int(0)
        cnt = 0
        # test_nv_dec's hot loop (one NAL per jm_nvdec_decode_frame call, fetch a frame whenever got_frame == 1) runs in the library:
        # a Python loop would measure the interpreter's per-call overhead and the GIL hand-off between the S feeder threads
        got_n = L.jm_amddec_feed_annexb(data, len(data), passes, C.cast(out, C.POINTER(C.c_ubyte)), frame_bytes, h)
        if got_n < 0:
            raise SystemExit("feed failed: " + L.jm_amddec_last_error(h).decode())
        cnt += got_n
        # let the pipeline run dry (no EOS: the handle keeps its DPB) and collect what it finished.
        # An access-unit delimiter carries no picture; two of them push the last slice NAL out of the
        # splitter (a NAL ends at the next start code) and close the picture (7.4.1.2.3).
        sc = b"\x00\x00\x01\x46\x01\x50" if args.codec == "hevc" else b"\x00\x00\x01\x09\x10"
        for _ in range(2):
            L.jm_amddec_decode_frame(C.cast(C.c_char_p(sc), C.c_void_p), len(sc), C.byref(got), h)
            if got.value == 1:
                n.value = frame_bytes
                if L.jm_amddec_output_frame(C.cast(out, C.c_void_p), C.byref(n), h) > 0:
                    cnt += 1
        L.jm_amddec_set_option(h, b"wait_idle", 1)
        while True:
            L.jm_amddec_decode_frame(C.cast(C.c_char_p(sc), C.c_void_p), len(sc), C.byref(got), h)
            if got.value != 1:
                break
            n.value = frame_bytes
            if L.jm_amddec_output_frame(C.cast(out, C.c_void_p), C.byref(n), h) > 0:
                cnt += 1
        counts[i] += cnt

    def batch(passes):
        ts = [threading.Thread(target=run_passes, args=(i, passes)) for i in range(S)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    def sync():
        if not args.parse_only and n_dev > 0:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    if W > 0:
        batch(W)
    KN = ("inter", "intra", "deblock", "packout")
    def eng():      # engine-wide counters (one engine per device serves every handle)
        return {k: {f: L.jm_amddec_get_stat(handles[0], f"k_{k}_{f}".encode()) for f in ("ns", "n", "pics", "alg_bytes")} for k in KN}
    e0 = eng()
    b0 = (L.jm_amddec_get_stat(handles[0], b"eng_batches"), L.jm_amddec_get_stat(handles[0], b"eng_batch_pics"))
    et0 = (L.jm_amddec_get_stat(handles[0], b"eng_launch_ns"), L.jm_amddec_get_stat(handles[0], b"eng_complete_ns"))
    jb0 = [L.jm_amddec_get_stat(h, b"job_bytes") for h in handles]
    pic0 = [L.jm_amddec_get_stat(h, b"pictures") for h in handles]
    for i in range(S):
        counts[i] = 0
    def host_cpu():     # CPU seconds of this process, and the container's CPU quota / throttling (cgroup v2), if visible
        import resource
        ru = resource.getrusage(resource.RUSAGE_SELF)
        d = {"cpu_s": ru.ru_utime + ru.ru_stime}
        try:
            for ln in open("/sys/fs/cgroup/cpu.stat"):
                k, v = ln.split()
                if k in ("nr_throttled", "throttled_usec"):
                    d[k] = int(v)
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            d["quota_cpus"] = None if q == "max" else round(int(q) / int(per), 2)
        except Exception:
            pass
        return d
    def thread_cpu():   # CPU seconds per thread name (user, system), from /proc
        tick = os.sysconf("SC_CLK_TCK"); acc = {}
        try:
            for tid in os.listdir("/proc/self/task"):
                f = open(f"/proc/self/task/{tid}/stat").read()
                name = f[f.index("(") + 1:f.rindex(")")]; rest = f[f.rindex(")") + 2:].split()
                u, sy = int(rest[11]) / tick, int(rest[12]) / tick
                a = acc.setdefault(name, [0.0, 0.0, 0]); a[0] += u; a[1] += sy; a[2] += 1
        except Exception:
            pass
        return acc
    sync()
    tc0 = thread_cpu()
    hc0 = host_cpu()
    t0 = time.perf_counter()
    batch(K)
    sync()
    dt = time.perf_counter() - t0
    hc1 = host_cpu()
    tc1 = thread_cpu()
    by_thread = {k: {"user_s": round(v[0] - tc0.get(k, [0, 0, 0])[0], 2), "sys_s": round(v[1] - tc0.get(k, [0, 0, 0])[1], 2), "threads": v[2]} for k, v in tc1.items()}
    by_thread = {k: v for k, v in by_thread.items() if v["user_s"] + v["sys_s"] >= 0.05}
    frames_local = sum(counts)

    from jmcodec_amd import shard
    frames_total, dt_max = shard.reduce_result(dist, frames_local, dt, device=red_dev)     # SUM of frames, MAX of seconds over ranks

    # ---- per-kernel device time: HIP events recorded by the engine on ITS stream around every batched launch, timed region only ----
    names = KN
    e1 = eng()
    tot_ns = {k: e1[k]["ns"] - e0[k]["ns"] for k in names}
    tot_n = {k: e1[k]["n"] - e0[k]["n"] for k in names}
    tot_pics = {k: e1[k]["pics"] - e0[k]["pics"] for k in names}
    tot_alg = {k: e1[k]["alg_bytes"] - e0[k]["alg_bytes"] for k in names}
    batches = L.jm_amddec_get_stat(handles[0], b"eng_batches") - b0[0]
    batch_pics = L.jm_amddec_get_stat(handles[0], b"eng_batch_pics") - b0[1]
    eng_thread_ms = {"launch_per_batch": round((L.jm_amddec_get_stat(handles[0], b"eng_launch_ns") - et0[0]) / 1e6 / max(batches, 1), 4), "retire_per_batch": round((L.jm_amddec_get_stat(handles[0], b"eng_complete_ns") - et0[1]) / 1e6 / max(batches, 1), 4)}
    job_bytes = sum(L.jm_amddec_get_stat(h, b"job_bytes") - jb0[i] for i, h in enumerate(handles))
    pictures = sum(L.jm_amddec_get_stat(h, b"pictures") - pic0[i] for i, h in enumerate(handles))
    errors = sum(L.jm_amddec_get_stat(h, b"errors") for h in handles)
    threads = L.jm_amddec_get_stat(handles[0], b"threads")
    host_diag = {k: round(sum(L.jm_amddec_get_stat(h, k.encode()) for h in handles) / 1e6 / max(1, sum(L.jm_amddec_get_stat(h, b"pictures") for h in handles)), 4)
                 for k in ("submit_ns", "wait_slot_ns", "parse_ns_i", "parse_ns_p")}   # ms per picture, whole run

    # algorithmic bytes per launch: summed by the engine over the pictures each batched launch processed
    surf = 1.5 * mb_w * 16 * mb_h * 16
    J = job_bytes / max(pictures, 1)
    p_frac = (F - F // 30) / F if F >= 30 else 1.0
    alg = {k: (tot_alg[k] / tot_n[k]) if tot_n[k] else 0.0 for k in names}
    # dominant kernel among the HBM-side stages; k_packout writes to pinned HOST memory, so it is priced against PCIe below
    dominant = max(("inter", "intra", "deblock"), key=lambda k: tot_ns[k])
    avg_s = {k: (tot_ns[k] * 1e-9 / tot_n[k]) if tot_n[k] else 0.0 for k in names}
    peak = 8000.0
    achieved = alg[dominant] / avg_s[dominant] / 1e9 if avg_s[dominant] > 0 else 0.0
    # HBM traffic from PMC counters (profiles/r01_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE passes, per picture) x pictures per launch
    traffic = None
    try:
        if args.codec == "hevc":
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_hevc_pmc_traffic.json")))["kernels"]
            t = lambda k: pmc[k]["traffic_upper"] if k in pmc else 0
            per_pic = {"inter": t("k_hevc_mc") + t("k_hevc_resid") + t("k_hevc_iresid"), "intra": t("k_hevc_intra"), "deblock": 2 * t("k_hevc_deblock") + t("k_hevc_sao")}
        else:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
            per_pic = {"inter": pmc["k_recon_inter"]["traffic_upper"], "intra": pmc.get("k_intra_band", pmc.get("k_intra_lds", {"traffic_upper": 0}))["traffic_upper"],
                       "deblock": pmc.get("k_deblock_band", pmc.get("k_deblock_lds", {"traffic_upper": 0}))["traffic_upper"] + pmc["k_deblock_prep"]["traffic_upper"]}
        if (args.width, args.height) == (1920, 1080):
            traffic = int(per_pic[dominant] * tot_pics[dominant] / max(tot_n[dominant], 1))
    except Exception:
        traffic = None
    # frame-level contract figure of SURVEY 8(d): A = 1.5*Wc*Hc*(n_ref+1) + 1.5*Wd*Hd + J per frame
    A = surf * (p_frac * 2 + (1 - p_frac) * 1) + frame_bytes + J
    kernel_s_per_frame = sum(tot_ns[k] for k in names) * 1e-9 / max(pictures, 1)

    # drain + tear down (outside the timed region)
    for h in handles:
        L.jm_amddec_decode_frame(None, 0, C.byref(C.c_int(0)), h)
        jmcodec_amd.jm_nvdec_deinit(h)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the build's own scalar CPU oracle (NOT libmfx: unobtainable),
        # one core, on a bounded sample of the same workload.
        o = streams.OracleHevc() if args.codec == "hevc" else streams.Oracle()
        sample_frames = min(F, 90 if args.codec == "h264" else 16)
        gen = streams.generate_hevc if args.codec == "hevc" else streams.generate
        sample = gen(**dict(cfg, frames=sample_frames)) if sample_frames != F else data
        reps, c0, n_dec = 0, time.perf_counter(), 0
        while True:
            _, n, _, _ = o.decode(sample, 1)
            n_dec += n
            reps += 1
            if time.perf_counter() - c0 > 10.0 or reps >= 8:
                break
        cdt = time.perf_counter() - c0
        cpu = {"value": round(n_dec / cdt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"{reps} x the first {sample_frames} frames of the same {args.width}x{args.height} stream, "
                         f"build CPU oracle (scalar, spec-literal; not libmfx), {cdt:.1f}s of CPU work",
               "host_cpus": os.cpu_count()}

    value = frames_total / dt_max
    line = {
        "metric": "decoded frames/sec @1080p H.264 + bit-exact YUV" if args.codec == "h264" else f"decoded frames/sec HEVC {args.width}x{args.height} (diagnostic, SURVEY 8d C3)",
        "value": round(value, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(dt_max * 1000.0 / K, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": (f"HEVC {tools_desc} + deblocking, IDR every 32, QP 32) " if args.codec == "hevc" else f"H.264 {tools_desc}, IDR every 30, QP 28, deblock on) ") + f"{args.width}x{args.height}, "
                               f"{S} independent streams per GPU x {F} frames per step, NAL-per-call via jm_nvdec_* API, I420 out",
                   "streams_per_gpu": S, "frames_per_stream_per_step": F, "bitstream_bytes": len(data),
                   "host_parse_threads": int(threads), "includes": "host entropy decode + H2D + kernels + packout + D2H into the caller's buffer (two of five handles: synchronous DMA from device staging; the others: pinned slot + memcpy)"},
        "frames": frames_total,
        "decode_errors": int(errors),
        "host_ms_per_picture": host_diag,
        "host_cpu": {"cpu_s": round(hc1["cpu_s"] - hc0["cpu_s"], 3), "cpus_busy": round((hc1["cpu_s"] - hc0["cpu_s"]) / dt, 2),
                     "cpu_ms_per_frame": round(1e3 * (hc1["cpu_s"] - hc0["cpu_s"]) / max(frames_local, 1), 4),
                     "quota_cpus": hc1.get("quota_cpus"), "online_cpus": os.cpu_count(),
                     "throttled_ms": round((hc1.get("throttled_usec", 0) - hc0.get("throttled_usec", 0)) / 1e3, 1), "by_thread": by_thread,
                     "note": "rank 0, timed region; when cpus_busy sits at quota_cpus the host half (entropy decode) bounds the rate"},
        "roofline": {"bound": "hbm", "kernel": "k_" + dominant, "achieved": round(achieved, 3), "peak": peak, "unit": "GB/s",
                     "frac": round(achieved / peak, 6), "traffic": traffic,
                     "alg_bytes_per_launch": int(alg[dominant]), "avg_launch_us": round(avg_s[dominant] * 1e6, 2),
                     "launches": int(tot_n[dominant]), "pictures_per_launch": round(tot_pics[dominant] / max(tot_n[dominant], 1), 2)},
        "engine": {"batches": int(batches), "pictures_per_batch": round(batch_pics / max(batches, 1), 2), "engine_thread_ms": eng_thread_ms},
        "pcie_out": {"bound": "pcie", "achieved": round(value / world * frame_bytes / 1e9, 2), "peak": 63.0, "unit": "GB/s",
                     "note": "tight frames: k_packout -> device staging, then copy engine -> caller's buffer (or -> pinned host slot ahead of time + memcpy); rate = frames/s x frame bytes per GPU"},
        "kernels": {("k_" + k): {"launches": int(tot_n[k]), "avg_us": round(avg_s[k] * 1e6, 2), "pictures_per_launch": round(tot_pics[k] / max(tot_n[k], 1), 2),
                                 "alg_GBps": round(alg[k] / avg_s[k] / 1e9, 2) if avg_s[k] > 0 else None} for k in names},
        "roofline_frame": {"alg_bytes_per_frame": int(A), "job_bytes_per_frame": int(J),
                           "end_to_end_GBps": round(value / world * A / 1e9, 3), "end_to_end_frac": round(value / world * A / 1e9 / peak, 6),
                           "kernel_time_GBps": round(A / kernel_s_per_frame / 1e9, 3) if kernel_s_per_frame > 0 else None},
    }
    if cpu is not None:
        line["cpu_baseline"] = cpu
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
