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


def quota_cpus():
    """CPUs this container may really use: the cgroup v2 CFS quota when there is one (a box can show 256 CPUs and own 16), else the online count."""
    cpus = float(os.cpu_count() or 8)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cpus = min(cpus, int(q) / int(per))
    except Exception:
        pass
    return cpus


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("JM_BENCH_STREAMS", "32")))
    ap.add_argument("--frames", type=int, default=60, help="frames per stream per step (multiple of the GOP, 30)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tools", default="baseline", choices=["baseline", "high", "high_b", "paff", "paff_b"],
                    help="diagnostic: coding tools of the synthetic stream (default = BASELINE config 1; high = CABAC + 8x8 transform; high_b = + I B B P; "
                         "paff = interlaced Main profile, every I / P picture a frame or two field pictures; paff_b = field pairs throughout, I B B P)")
    ap.add_argument("--codec", default="h264", choices=["h264", "hevc"], help="diagnostic: hevc = SURVEY 8d config C3 (HEVC Main, 64x64 CTB, SAO + "
        "deblocking, random-access GOP 8) at --width x --height")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the untimed single-stream leg")
    ap.add_argument("--no-profile", action="store_true", help="diagnostic: do not record per-kernel HIP events")
    ap.add_argument("--device-output", action="store_true", help="diagnostic (profiles/): display frames stay in device memory "
        "(jm_amddec_output_frame_device), no D2H copy -- "
                    "under rocprofv3 the runtime replaces copy-engine transfers by blit kernels, which perturbs the decode kernels")
    ap.add_argument("--parse-only", action="store_true", help="diagnostic: host stages only (no device work, frames carry no pixels)")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed extra legs (c4_slice, c2_4k, c3_4k) that the default invocation appends")
    args = ap.parse_args()

    ctx = setup_process(args)
    line = measure(args, ctx)
    dist = ctx["dist"]
    if ctx["rank"] == 0 and line is not None:
        # ---- untimed extra legs: the other BASELINE configurations, driver-visible.  The headline line and its timed region above
        # are unchanged; each leg is a full measure() of its own (fresh handles, warm-up, timed steps, bit-exact check against the oracle) reduced to a
        # compact object.  Only in the default invocation (N = 1, the headline workload): diagnostics and multi-rank runs skip them. ----
        if ctx["world"] == 1 and default_workload(args) and not args.no_extra:
            # (the measured headline must survive whatever a leg does -- it goes to stderr before the legs start, and a leg that raises ANYTHING
            # becomes an {"error": ...} entry of the one JSON line instead of taking the line with it)
            print("bench.py: headline before the extra legs (repeated in the final line): " +
                  json.dumps({k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "bit_exact") if k in line}), file=sys.stderr, flush=True)
            for key, over in EXTRA_LEGS:
                t0 = time.perf_counter()
                try:
                    leg = measure(leg_args(args, over), ctx)
                    line[key] = compact_leg(leg, time.perf_counter() - t0)
                except (SystemExit, Exception) as e:      # a leg that cannot run (memory, a failed init, a bug in the leg's own code)
                    line[key] = {"error": f"{type(e).__name__}: {e}"}
                    print(f"bench.py: extra leg {key} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if line is not None and (line.get("bit_exact") is False or any(isinstance(line.get(k), dict) and line[k].get("bit_exact") is False for k, _ in EXTRA_LEGS)):
        raise SystemExit(3)
    if line is None and ctx.get("bit_exact_local") is False:
        raise SystemExit(3)


# BASELINE.json configs[2..4] at their sizes (SURVEY 8d C2, C3, and the per-GPU slice of C4); sizes per VERDICT r3 next 1b
EXTRA_LEGS = (
    ("c4_slice", dict(streams=8, frames=60, steps=10, warmup=1)),
    ("c2_4k", dict(tools="high_b", width=3840, height=2160, streams=16, frames=24, steps=3, warmup=1)),
    ("c3_4k", dict(codec="hevc", width=3840, height=2160, streams=16, frames=16, steps=3, warmup=1)),
)


def default_workload(args):
    if os.environ.get("JM_BENCH_TEST_LEGS"):      # tests (no GPU): the legs at a toy size through --parse-only, see leg_args
        return True
    return (args.codec == "h264" and args.tools == "baseline" and args.width == 1920 and args.height == 1080 and args.streams == 32 and
            not args.device_output and not args.parse_only)


def leg_args(args, over):
    import copy
    a = copy.copy(args)
    a.no_cpu_baseline = True; a.no_single = True; a.leg = True
    for k, v in over.items():
        setattr(a, k, v)
    if os.environ.get("JM_BENCH_TEST_LEGS"):      # "WxH": the legs' code path on a host without a GPU (tests/test_sharding.py)
        a.width, a.height = (int(v) for v in os.environ["JM_BENCH_TEST_LEGS"].split("x"))
        a.streams, a.steps = min(a.streams, 3), 1
    return a


def compact_leg(l, wall_s):
    r = l["roofline"]
    return {"value": l["value"], "unit": l["unit"], "workload": l["config"]["workload"], "steps": l["steps"], "ms_per_step": l["ms_per_step"],
            "bit_exact": l["bit_exact"], "frames_checked": l["frames_checked"], "decode_errors": l["decode_errors"], "scaling_bound": l["scaling_bound"],
            "roofline": {k: r[k] for k in ("kernel", "frac", "achieved", "traffic", "traffic_raw", "alg_bytes_per_launch", "avg_launch_us",
                                             "pictures_per_launch")},
            "kernels": {k: {"avg_us": v["avg_us"], "pictures_per_launch": v["pictures_per_launch"]} for k, v in l["kernels"].items() if v["launches"]},
            "host_cpu": {"cpu_ms_per_frame": l["host_cpu"]["cpu_ms_per_frame"], "cpus_busy": l["host_cpu"]["cpus_busy"]},
            "engine": {k: l["engine"][k] for k in ("chain_batches", "chain_i_batches", "chain_recoveries", "device_wait_errors", "pictures_per_batch",
                                                      "chain_launches_with_clock_gaps")},
            "stream_generation_s": l["config"]["stream_generation_s"], "leg_wall_s": round(wall_s, 1)}


def setup_process(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    dist = None
    # north_star: "independent input streams shard one-per-GPU ... no RCCL" -- the only cross-rank traffic is the barrier and two scalars, so the
    # process group is gloo (CPU) by default; JM_BENCH_BACKEND=nccl is kept only as a diagnostic
    backend = os.environ.get("JM_BENCH_BACKEND", "gloo")
    n_dev = torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    os.environ["JM_AMD_DEC_DEVICE"] = str(local_rank % max(n_dev, 1))
    if world > 1 and "JM_AMD_DEC_THREADS" not in os.environ:      # the ranks of a node share its host cores (parse workers per rank)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        os.environ["JM_AMD_DEC_THREADS"] = str(max(4, min(64, int(quota_cpus() * 1.25 + 0.5) // max(local_world, 1))))
    red_dev = "cuda" if (world > 1 and backend == "nccl") else "cpu"
    return {"rank": rank, "local_rank": local_rank, "world": world, "dist": dist, "torch": torch, "n_dev": n_dev, "red_dev": red_dev}


def measure(args, ctx):
    """One configuration: handles, warm-up, K timed steps, the untimed bit-exact pass; returns the result line (rank 0) or None."""
    rank, local_rank, world, dist, torch, n_dev, red_dev = (ctx[k] for k in ("rank", "local_rank", "world", "dist", "torch", "n_dev", "red_dev"))

    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "jmcodec_amd", "lib", "libjm_amd_dec.so")):
        ge.build()
    import jmcodec_amd
    from tools import streams
    L = jmcodec_amd.lib()
    if not jmcodec_amd.jm_nvdec_is_hw_support() and not args.parse_only:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")

    # ---- synthetic input: S DISTINCT streams per rank.  SURVEY 8d C1 / C4: seed = 0x4A4D0000 + 1*256 + stream_id, and the job's S*world streams
    # shard as stream i -> rank i mod world (jmcodec_amd/shard.py) ----
    from jmcodec_amd import shard
    S, F, K, W = args.streams, args.frames, args.steps, args.warmup
    stream_ids = shard.streams_of_rank(S * world, rank, world)
    assert len(stream_ids) == S
    tools_desc = "Baseline, I/P-only (CAVLC"
    if args.tools != "baseline":
        tools_desc = "High, I/P-only (CABAC, 8x8 transform"
    if args.tools == "high_b":
        tools_desc = "High, I B B P (CABAC, 8x8 transform"
    if args.tools == "paff":
        tools_desc = "Main interlaced, I/P frame or field-pair pictures (PAFF, CABAC"
    if args.tools == "paff_b":
        tools_desc = "Main interlaced, field pairs, I B B P (PAFF, CABAC"
    if args.codec == "hevc":
        tools_desc = "Main, random-access GOP 8 (CABAC, 64x64 CTB, SAO"

    def stream_cfg(sid, frames=None):
        if args.codec == "hevc":
            return streams.config_c3(frames=frames or F, width=args.width, height=args.height, stream_id=sid)
        cfg = streams.config_c1(stream_id=sid, frames=frames or F, width=args.width, height=args.height)
        if args.tools in ("high", "high_b"):
            cfg.update(cabac=1, t8x8=1)
        if args.tools == "high_b":
            cfg.update(bframes=2, num_ref=2, poc_type=0)
        if args.tools == "paff":
            cfg.update(cabac=1, paff=1, num_ref=2, poc_type=0)
        if args.tools == "paff_b":
            cfg.update(cabac=1, paff=2, bframes=2, num_ref=2, poc_type=0)
        return cfg

    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    host_threads = max(2, min(32, int(quota_cpus() // max(local_world, 1))))

    def make_stream(sid):
        """Generated once per box and cached under /tmp (the 1/2/4/8-GPU runs of a scaling sweep share most of their streams)."""
        import hashlib
        cfg = stream_cfg(sid)
        key = hashlib.md5(repr((args.codec, sorted(cfg.items()), os.path.getmtime(os.path.join(ROOT, "tools",
            "hevcgen.c" if args.codec == "hevc" else "h264gen.c")))).encode()).hexdigest()
        path = os.path.join(os.environ.get("JM_BENCH_CACHE", "/tmp"), f"jm_bench_{key}.bin")
        try:
            with open(path, "rb") as f:
                return f.read()
        except OSError:
            pass
        d = (streams.generate_hevc if args.codec == "hevc" else streams.generate)(**cfg)
        try:
            tmp = path + f".{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(d)
            os.replace(tmp, path)
        except OSError:
            pass
        return d

    from concurrent.futures import ThreadPoolExecutor      # the generator / oracle are C libraries called through ctypes: the GIL is released
    tg0 = time.perf_counter()
    with ThreadPoolExecutor(host_threads) as ex:
        datas = list(ex.map(make_stream, stream_ids))
    gen_s = time.perf_counter() - tg0
    mb_w, mb_h = (args.width + 15) // 16, (args.height + 15) // 16
    frame_bytes = args.width * args.height * 3 // 2

    # memory the ranks of this node will want (measured: 0.2 GB of page-locked job buffers per 1080p H.264 handle, x4 at 4K, + ~2 GB of runtime and Python
    # per rank, + the oracle check's transient) against the container's allotment: say so BEFORE the kernel's OOM killer does
    try:
        mm = open("/sys/fs/cgroup/memory.max").read().strip()
        if mm != "max" and not args.parse_only:
            need = local_world * (S * 0.2e9 * max(1.0, args.width * args.height / (1920.0 * 1080.0)) + 3e9)
            if need > 0.9 * int(mm):
                print(f"bench.py: WARNING: {local_world} rank(s) x {S} handles want about {need / 1e9:.0f} GB of host memory, the container allows "
                    f"{int(mm) / 1e9:.0f} GB", file=sys.stderr, flush=True)
    except (OSError, ValueError):
        pass
    handles = []
    for _ in range(S):
        h = jmcodec_amd.jm_nvdec_create_handle()
        L.jm_amddec_set_option(h, b"profile", 0 if args.no_profile else 1)   # the engine records HIP events around each batched launch
        if args.parse_only:
            L.jm_amddec_set_option(h, b"parse_only", 1)
        if args.device_output:
            L.jm_amddec_set_option(h, b"device_output", 1)
        if jmcodec_amd.jm_nvdec_init(1 if args.codec == "hevc" else 0, 1, None, 0, h) != 0:
            raise SystemExit("init failed: " + L.jm_amddec_last_error(h).decode())
        handles.append(h)

    counts = [0] * S

    dev_ptr, dev_len = C.c_void_p(0), C.c_int(0)

    def take(out, n, h):
        """jm_nvdec_output_frame -- or, with --device-output, the device-resident variant (no copy)."""
        if args.device_output:
            p, ln = C.c_void_p(0), C.c_int(0)
            return L.jm_amddec_output_frame_device(C.byref(p), C.byref(ln), h)
        n.value = frame_bytes
        return L.jm_amddec_output_frame(C.cast(out, C.c_void_p), C.byref(n), h)

    feeder_cpu = [0.0] * S          # CPU seconds of each handle's calling thread (they exit before the per-thread /proc snapshot)
    out_bufs = [C.create_string_buffer(frame_bytes) for _ in range(S)]

    def run_passes(i, passes):
        """test_nv_dec's hot loop: one NAL per jm_nvdec_decode_frame call, pull a frame whenever got_frame == 1."""
        tc_start = time.thread_time()
        h = handles[i]
        # one output buffer per handle for the whole run, as the reference harness has (test_nv_dec.cpp:107-110).  JM_BENCH_FRESH_BUFFERS=1: a new one per
        # pass, freed when the pass ends (rounds 1-5; the munmap of a buffer that the output route had page-locked is suspected of taking the process's
        # queues off the device for ~25 ms: profiles/r06_chain_soak.txt)
        out = C.create_string_buffer(frame_bytes) if os.environ.get("JM_BENCH_FRESH_BUFFERS") else out_bufs[i]
        got = C.c_int(0)
        n = C.c_int(0)
        cnt = 0
        # test_nv_dec's hot loop (one NAL per jm_nvdec_decode_frame call, fetch a frame whenever got_frame == 1) runs in the library:
        # a Python loop would measure the interpreter's per-call overhead and the GIL hand-off between the S feeder threads
        data = datas[i]
        got_n = L.jm_amddec_feed_annexb(data, len(data), passes, None if args.device_output else C.cast(out, C.POINTER(C.c_ubyte)), frame_bytes, h)
        if got_n < 0:
            raise SystemExit("feed failed: " + L.jm_amddec_last_error(h).decode())
        cnt += got_n
        # let the pipeline run dry (no EOS: the handle keeps its DPB) and collect what it finished.
        # An access-unit delimiter carries no picture; two of them push the last slice NAL out of the
        # splitter (a NAL ends at the next start code) and close the picture (7.4.1.2.3).
        sc = b"\x00\x00\x01\x46\x01\x50" if args.codec == "hevc" else b"\x00\x00\x01\x09\x10"
        for _ in range(2):
            L.jm_amddec_decode_frame(C.cast(C.c_char_p(sc), C.c_void_p), len(sc), C.byref(got), h)
            if got.value == 1 and take(out, n, h) > 0:
                cnt += 1
        L.jm_amddec_set_option(h, b"wait_idle", 1)
        while True:
            L.jm_amddec_decode_frame(C.cast(C.c_char_p(sc), C.c_void_p), len(sc), C.byref(got), h)
            if got.value != 1:
                break
            if take(out, n, h) > 0:
                cnt += 1
        counts[i] += cnt
        feeder_cpu[i] += time.thread_time() - tc_start

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

    CS = (b"eng_chain_batches", b"eng_chain_pics", b"eng_wait_errors", b"eng_chain_recoveries", b"eng_chain_i_batches", b"eng_wait_gap_launches")
    cs0 = [L.jm_amddec_get_stat(handles[0], k) for k in CS]      # the engine is process-wide: counters of THIS measure() = differences
    if W > 0:
        batch(W)
    KN = ("inter", "intra", "deblock", "packout", "chain")     # chain = k_chain: reconstruction + deblocking of consecutive pictures in one launch
    def eng():      # engine-wide counters (one engine per device serves every handle)
        return {k: {f: L.jm_amddec_get_stat(handles[0], f"k_{k}_{f}".encode()) for f in ("ns", "n", "pics", "alg_bytes")} for k in KN}
    LK = [b"eng_rej_other_lane", b"eng_rej_cross_lane", b"eng_rej_tables", b"eng_early_intra", b"eng_blocked_ns", b"eng_blocked_n", b"eng_forms"] + \
         [f"eng_lane{i}_{f}".encode() for i in range(4) for f in ("busy_ns", "gap_ns", "batches", "pics", "upwait_ns", "prewait_ns", "dry")]
    def lanes_now(h=None):
        return {k.decode(): L.jm_amddec_get_stat(h or handles[0], k) for k in LK}
    def lanes_report(a, b_, wall_s):
        """Engine lanes between two snapshots: how full the ordinary lane's batches were and why not fuller, and how busy each lane's stream was."""
        d = {k: b_[k] - a[k] for k in a}
        forms = max(d["eng_forms"], 1)
        names = ("ordinary", "intra", "hevc", "hevc_intra")
        out = {"left_out_per_ordinary_batch": {"oldest_picture_is_for_another_lane": round(d["eng_rej_other_lane"] / forms, 2),
                                               "earlier_pictures_on_another_lane": round(d["eng_rej_cross_lane"] / forms, 2),
                                               "pack_tables_full": round(d["eng_rej_tables"] / forms, 2)},
               "intra_pictures_launched_ahead_of_their_turn": int(d["eng_early_intra"]),
               "left_out_ms_per_occasion": round(d["eng_blocked_ns"] / 1e6 / max(d["eng_blocked_n"], 1), 3), "left_out_occasions": int(d["eng_blocked_n"])}
        for i, nm in enumerate(names):
            if d[f"eng_lane{i}_batches"]:
                out[nm] = {"batches": int(d[f"eng_lane{i}_batches"]), "pictures_per_batch": round(d[f"eng_lane{i}_pics"] / d[f"eng_lane{i}_batches"], 2),
                           "busy_frac": round(d[f"eng_lane{i}_busy_ns"] / 1e9 / wall_s, 3), "idle_between_batches_frac": round(d[f"eng_lane{i}_gap_ns"] / 1e9 / wall_s, 3),
                           "kernel_ms_per_batch": round(d[f"eng_lane{i}_busy_ns"] / 1e6 / d[f"eng_lane{i}_batches"], 3),
                           # where the idle time went (H.264 lanes): the batch's job lists had not landed / its pre-pass had not finished when the previous
                           # batch ended (fractions of wall time, the second includes the first); batches launched after the previous one had already ended
                           "idle_waiting_for_job_lists_frac": round(d[f"eng_lane{i}_upwait_ns"] / 1e9 / wall_s, 3),
                           "idle_waiting_for_pre_pass_frac": round(d[f"eng_lane{i}_prewait_ns"] / 1e9 / wall_s, 3),
                           "batches_launched_after_the_lane_ran_dry": int(d[f"eng_lane{i}_dry"])}
        return out
    e0 = eng()
    ln0 = lanes_now()
    b0 = (L.jm_amddec_get_stat(handles[0], b"eng_batches"), L.jm_amddec_get_stat(handles[0], b"eng_batch_pics"))
    et0 = (L.jm_amddec_get_stat(handles[0], b"eng_launch_ns"), L.jm_amddec_get_stat(handles[0], b"eng_complete_ns"))
    fm0 = [L.jm_amddec_get_stat(handles[0], k) for k in (b"eng_forms", b"eng_form_decoders", b"eng_form_pending")]
    jb0 = [L.jm_amddec_get_stat(h, b"job_bytes") for h in handles]
    pic0 = [L.jm_amddec_get_stat(h, b"pictures") for h in handles]
    for i in range(S):
        counts[i] = 0
        feeder_cpu[i] = 0.0
    def memtrace(tag):   # JM_BENCH_MEMTRACE=1: resident / locked memory of this rank at the phases of the run (stderr)
        if not os.environ.get("JM_BENCH_MEMTRACE"):
            return
        try:
            f = {l.split(":")[0]: l.split(":")[1].strip() for l in open("/proc/self/status") if l.startswith(("VmRSS", "VmHWM", "VmLck", "VmPin", "RssAnon",
                "RssShmem", "RssFile"))}
            print(f"bench.py: memtrace rank {rank} {tag}: {f}", file=sys.stderr, flush=True)
        except OSError:
            pass

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
    memtrace("handles created, warm-up done")
    tc0 = thread_cpu()
    hc0 = host_cpu()
    t0 = time.perf_counter()
    batch(K)
    sync()
    dt = time.perf_counter() - t0
    hc1 = host_cpu()
    tc1 = thread_cpu()
    memtrace("timed region done")
    import resource as _res
    peak_rss_timed_mb = round(_res.getrusage(_res.RUSAGE_SELF).ru_maxrss / 1024.0, 1)     # high-water mark up to here: handles, streams, the timed passes
    by_thread = {k: {"user_s": round(v[0] - tc0.get(k, [0, 0, 0])[0], 2), "sys_s": round(v[1] - tc0.get(k, [0, 0, 0])[1], 2), "threads": v[2]} for k,
        v in tc1.items()}
    by_thread = {k: v for k, v in by_thread.items() if v["user_s"] + v["sys_s"] >= 0.05}
    frames_local = sum(counts)

    frames_total, dt_max = shard.reduce_result(dist, frames_local, dt, device=red_dev)     # SUM of frames, MAX of seconds over ranks

    # ---- per-kernel device time: HIP events recorded by the engine on ITS stream around every batched launch, timed region only ----
    names = KN
    e1 = eng()
    lanes_timed = lanes_report(ln0, lanes_now(), dt)
    tot_ns = {k: e1[k]["ns"] - e0[k]["ns"] for k in names}
    tot_n = {k: e1[k]["n"] - e0[k]["n"] for k in names}
    tot_pics = {k: e1[k]["pics"] - e0[k]["pics"] for k in names}
    tot_alg = {k: e1[k]["alg_bytes"] - e0[k]["alg_bytes"] for k in names}
    batches = L.jm_amddec_get_stat(handles[0], b"eng_batches") - b0[0]
    batch_pics = L.jm_amddec_get_stat(handles[0], b"eng_batch_pics") - b0[1]
    cs1 = [L.jm_amddec_get_stat(handles[0], k) for k in CS]
    gap_max_us = L.jm_amddec_get_stat(handles[0], b"eng_wait_gap_max_us")
    chain_stat = (cs1[0] - cs0[0], cs1[1] - cs0[1], cs1[2] - cs0[2], cs1[3] - cs0[3], L.jm_amddec_get_stat(handles[0], b"eng_gpu_shared"),
                  cs1[4] - cs0[4], cs1[5] - cs0[5])   # warm-up + timed region of this configuration
    dfr = sum(L.jm_amddec_get_stat(h, b"direct_frames") for h in handles)
    direct_stat = {"sdma_engines": hex(L.jm_amddec_get_stat(handles[0], b"copy_engines")), "frames_whole_run": int(dfr),
        "caller_wait_us_per_frame": round(sum(L.jm_amddec_get_stat(h, b"direct_ns") for h in handles) / 1e3 / max(dfr, 1), 1)}
    fm1 = [L.jm_amddec_get_stat(handles[0], k) for k in (b"eng_forms", b"eng_form_decoders", b"eng_form_pending")]
    forms = max(1, fm1[0] - fm0[0])
    form_stat = {"decoders_waiting_per_batch_formed": round((fm1[1] - fm0[1]) / forms, 2),
        "pictures_waiting_per_batch_formed": round((fm1[2] - fm0[2]) / forms, 2)}   # timed region, ordinary lane
    eng_thread_ms = {"launch_per_batch": round((L.jm_amddec_get_stat(handles[0], b"eng_launch_ns") - et0[0]) / 1e6 / max(batches, 1), 4),
        "retire_per_batch": round((L.jm_amddec_get_stat(handles[0], b"eng_complete_ns") - et0[1]) / 1e6 / max(batches, 1), 4)}
    job_bytes = sum(L.jm_amddec_get_stat(h, b"job_bytes") - jb0[i] for i, h in enumerate(handles))
    pictures = sum(L.jm_amddec_get_stat(h, b"pictures") - pic0[i] for i, h in enumerate(handles))
    threads = L.jm_amddec_get_stat(handles[0], b"threads")
    host_diag = {k: round(sum(L.jm_amddec_get_stat(h, k.encode()) for h in handles) / 1e6 / max(1, sum(L.jm_amddec_get_stat(h,
        b"pictures") for h in handles)), 4)
                 for k in ("submit_ns", "wait_slot_ns", "parse_ns_i", "parse_ns_p")}   # ms per picture, whole run

    # algorithmic bytes per launch: summed by the engine over the pictures each batched launch processed
    surf = 1.5 * mb_w * 16 * mb_h * 16
    J = job_bytes / max(pictures, 1)
    p_frac = (F - F // 30) / F if F >= 30 else 1.0
    alg = {k: (tot_alg[k] / tot_n[k]) if tot_n[k] else 0.0 for k in names}
    # dominant kernel among the HBM-side stages; k_packout writes to pinned HOST memory, so it is priced against PCIe below
    dominant = max(("inter", "intra", "deblock", "chain"), key=lambda k: tot_ns[k])
    avg_s = {k: (tot_ns[k] * 1e-9 / tot_n[k]) if tot_n[k] else 0.0 for k in names}
    peak = 8000.0
    achieved = alg[dominant] / avg_s[dominant] / 1e9 if avg_s[dominant] > 0 else 0.0
    # HBM traffic from PMC counters: one file per codec / tool set / picture size under profiles/ (separate FETCH_SIZE / WRITE_SIZE passes with ONE
    # stream, so bytes are per picture; tools/make_traffic_profile.py), scaled by the pictures per launch of THIS run.  No file for the size: null.
    traffic = traffic_raw = None
    traffic_file = None
    try:
        tag = f"hevc_{args.width}x{args.height}" if args.codec == "hevc" else f"h264_{args.tools}_{args.width}x{args.height}"
        cands = [f"r06_pmc_traffic_{tag}.json", f"r05_pmc_traffic_{tag}.json", f"r04_pmc_traffic_{tag}.json", f"r03_pmc_traffic_{tag}.json"]
        if tag == "h264_baseline_1920x1080":
            cands.append("r02_pmc_traffic.json")
        pmc_path = next(p for p in (os.path.join(ROOT, "profiles", f) for f in cands) if os.path.exists(p))
        traffic_file = os.path.relpath(pmc_path, ROOT)
        pmc = json.load(open(pmc_path))["kernels"]
        t = lambda k: pmc[k]["traffic_upper"] if k in pmc else 0
        r = lambda k: (pmc[k]["fetch_raw"] + pmc[k]["write"]) if k in pmc else 0
        if args.codec == "hevc":
            # k_hevc_deblock runs twice per picture (vertical edges, horizontal edges); the file holds the average of its launches
            per_pic = {"inter": t("k_hevc_mc") + t("k_hevc_resid") + t("k_hevc_iresid"), "intra": t("k_hevc_intra"),
                "deblock": 2 * t("k_hevc_deblock") + t("k_hevc_sao")}
            raw_pic = {"inter": r("k_hevc_mc") + r("k_hevc_resid") + r("k_hevc_iresid"), "intra": r("k_hevc_intra"),
                "deblock": 2 * r("k_hevc_deblock") + r("k_hevc_sao")}
        else:
            per_pic = {"inter": t("k_recon_inter"), "intra": t("k_intra_band"), "deblock": t("k_deblock_band") + t("k_deblock_prep")}
            raw_pic = {"inter": r("k_recon_inter"), "intra": r("k_intra_band"), "deblock": r("k_deblock_band") + r("k_deblock_prep")}
            per_pic["chain"] = pmc["k_chain"]["traffic_upper"] if "k_chain" in pmc else per_pic["inter"] + per_pic["deblock"]
            raw_pic["chain"] = r("k_chain") if "k_chain" in pmc else raw_pic["inter"] + raw_pic["deblock"]
        if per_pic.get(dominant):
            traffic = int(per_pic[dominant] * tot_pics[dominant] / max(tot_n[dominant], 1))
            traffic_raw = int(raw_pic[dominant] * tot_pics[dominant] / max(tot_n[dominant], 1))
    except (StopIteration, OSError, KeyError, ValueError):
        traffic = traffic_raw = None
    # frame-level contract figure of SURVEY 8(d): A = 1.5*Wc*Hc*(n_ref+1) + 1.5*Wd*Hd + J per frame
    A = surf * (p_frac * 2 + (1 - p_frac) * 1) + frame_bytes + J
    kernel_s_per_frame = sum(tot_ns[k] for k in names) * 1e-9 / max(pictures, 1)

    # ---- untimed: prove "+ bit-exact YUV" on the TIMED CONFIGURATION.  One more pass over every handle, all S handles concurrently exactly as in
    # the timed region (same engine batching, same mix of output routes), every frame MD5'd; the handles are then flushed and closed.  The digests
    # are compared with the CPU oracle (oracle/, checker only) decoding the first IDR period of the same stream: every handle at N=1, two per
    # rank otherwise (N ranks share the node's host cores). ----
    import hashlib
    is_hevc = args.codec == "hevc"

    idr_period = lambda data, which=0: streams.idr_period(data, which, is_hevc)

    def first_period(data):
        return idr_period(data, 0)

    check_digests = [None] * S

    def check_pass(i):
        h = handles[i]
        out = C.create_string_buffer(frame_bytes)
        got, n = C.c_int(0), C.c_int(0)
        digs = []

        def pull():
            n.value = frame_bytes
            if L.jm_amddec_output_frame(C.cast(out, C.c_void_p), C.byref(n), h) > 0:
                digs.append(hashlib.md5(out.raw[:n.value]).digest())
        for nal in jmcodec_amd.split_nalus(datas[i]):
            L.jm_amddec_decode_frame(C.cast(C.c_char_p(nal), C.c_void_p), len(nal), C.byref(got), h)
            if got.value == 1:
                pull()
        while not L.jm_amddec_is_exit(h):                      # EOS: flush what the DPB still holds, then the handle reports is_exit
            if L.jm_amddec_decode_frame(None, 0, C.byref(got), h) != 0:
                break
            if got.value == 1:
                pull()
        check_digests[i] = digs[-F:]                           # frames a reordering DPB still held from the timed passes come out first

    bit_exact, frames_checked, check_note = None, 0, "skipped (--parse-only)"
    oracle_dt, oracle_frames, oracle_one = 0.0, 0, None
    if not args.parse_only:
        ts = [threading.Thread(target=check_pass, args=(i,)) for i in range(S)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        orc = streams.OracleHevc() if is_hevc else streams.Oracle()
        n_oracle = S if world == 1 else min(S, 2)
        prefixes = [first_period(datas[i]) for i in range(n_oracle)]

        def oracle_digests(pfx):
            yuv, n, w, h = orc.decode(pfx, 1)
            fb = w * h * 3 // 2
            return [hashlib.md5(yuv[k * fb:(k + 1) * fb]).digest() for k in range(n)]
        want = [None] * n_oracle
        first = 0
        if world == 1 and not args.no_cpu_baseline:          # B1: one stream alone on one core
            c0 = time.perf_counter()
            want[0] = oracle_digests(prefixes[0])
            oracle_one = (len(want[0]), time.perf_counter() - c0)
            first = 1
        c0 = time.perf_counter()
        with ThreadPoolExecutor(host_threads) as ex:          # B2: the remaining streams on all the cores this container owns
            for k, d in zip(range(first, n_oracle), ex.map(oracle_digests, prefixes[first:])):
                want[k] = d
        oracle_dt = time.perf_counter() - c0
        oracle_frames = sum(len(w) for w in want[first:])
        bit_exact = all(len(check_digests[i]) == F for i in range(S))
        for i in range(n_oracle):
            ok = len(want[i]) > 0 and check_digests[i][:len(want[i])] == want[i]
            bit_exact = bit_exact and ok
            frames_checked += len(want[i])
            if not ok:
                bad = next((k for k in range(min(len(want[i]), len(check_digests[i]))) if want[i][k] != check_digests[i][k]), -1)
                print(f"bench.py: rank {rank} stream {stream_ids[i]}: decoded frames differ from the CPU oracle (first bad frame {bad}, got "
                    f"{len(check_digests[i])} frames, oracle {len(want[i])})", file=sys.stderr)
        # a LATE IDR period too (the last one of the pass: frames F - period .. F - 1) of up to two streams: everything between the first period and the end
        # of a pass -- DPB reuse over many pictures, surfaces and job slots recycled, chain launches in steady state -- is otherwise only self-consistent
        n_late, late_note = 0, ""
        n_periods = 0
        while idr_period(datas[0], n_periods) is not None:
            n_periods += 1
        if n_periods >= 2:
            for i in range(min(2, n_oracle)):
                late = oracle_digests(idr_period(datas[i], n_periods - 1))
                got_late = check_digests[i][F - len(late):] if len(late) else []
                ok = len(late) > 0 and got_late == late
                bit_exact = bit_exact and ok
                frames_checked += len(late); n_late += 1
                if not ok:
                    print(f"bench.py: rank {rank} stream {stream_ids[i]}: IDR period {n_periods - 1} differs from the CPU oracle", file=sys.stderr)
            late_note = f"; the last IDR period (period {n_periods - 1}, frames {F - len(late)}..{F - 1}) of {n_late} handle(s) as well"
        check_note = (f"one extra pass of all {S} handles concurrently (same batching and output routes as the timed passes), every frame MD5'd; "
                      f"first IDR period ({len(want[0])} frames) of {n_oracle} handle(s) per rank compared with the CPU oracle{late_note}; all {S} handles "
                      f"returned {F} frames")
    memtrace("check pass and oracle done")
    errors = sum(L.jm_amddec_get_stat(h, b"errors") for h in handles)
    job_slot_mb = round(sum(L.jm_amddec_get_stat(h, b"job_slot_bytes") for h in handles) / 1048576.0, 1)      # page-locked job buffers of all handles, as grown
    job_regrown = int(sum(L.jm_amddec_get_stat(h, b"job_regrown") for h in handles))
    numa_node = int(L.jm_amddec_get_stat(handles[0], b"numa_node"))
    for i, h in enumerate(handles):
        if L.jm_amddec_get_stat(h, b"errors"):
            print(f"bench.py: rank {rank} stream {stream_ids[i]}: {L.jm_amddec_get_stat(h, b'errors')} decode error(s), last: "
                f"{L.jm_amddec_last_error(h).decode()!r}", file=sys.stderr)
        jmcodec_amd.jm_nvdec_deinit(h)

    # ---- untimed: ONE handle fed exactly like test_nv_dec.cpp:184-250 (the reference harness's own shape: one stream, one thread) ----
    memtrace("handles released")
    single = None
    if world == 1 and not args.parse_only and not args.no_single:
        h = jmcodec_amd.jm_nvdec_create_handle()
        if jmcodec_amd.jm_nvdec_init(1 if is_hevc else 0, 1, None, 0, h) == 0:
            out = C.create_string_buffer(frame_bytes)
            sp = max(1, min(8, 240 // max(F, 1)))
            L.jm_amddec_feed_annexb(datas[0], len(datas[0]), 1, C.cast(out, C.POINTER(C.c_ubyte)), frame_bytes, h)
            L.jm_amddec_set_option(h, b"wait_idle", 1)
            c0 = time.perf_counter()
            n1 = L.jm_amddec_feed_annexb(datas[0], len(datas[0]), sp, C.cast(out, C.POINTER(C.c_ubyte)), frame_bytes, h)
            L.jm_amddec_set_option(h, b"wait_idle", 1)
            sdt = time.perf_counter() - c0
            single = {"value": round(sp * F / sdt, 1), "unit": "frames/s", "frames": sp * F, "frames_returned_in_loop": int(n1),
                      "note": "one jm_nvdec handle, one feeder thread, NAL-per-call, frame copied into the caller's buffer whenever got_frame == 1 "
                      "(test_nv_dec.cpp:184-250); untimed extra leg"}
        jmcodec_amd.jm_nvdec_deinit(h)

    # ---- untimed: BASELINE config 0's call shape -- ONE handle of the push / pull API (jm_intel_dec_*), driven by the loop of test_intel_dec.cpp:64-102
    # in native code (jm_amdintel_run_pushpull): input_data in pushes of free_buf_len = 1 MB while need_more_data, one output_frame into the caller's
    # buffer per turn, set_eof when the input ran out, until is_exit.  A first copy of the stream is pushed through the same handle untimed (allocations,
    # first launches), like the single-stream leg. ----
    c0_leg = None
    if world == 1 and not args.parse_only and not args.no_single:
        h = L.jm_amdintel_create_handle()
        if L.jm_amdintel_init(1 if is_hevc else 0, 1, h) == 0:
            out = (C.c_ubyte * frame_bytes)()
            sp = max(1, min(8, 240 // max(F, 1)))
            d0 = datas[0]
            base = C.cast(C.c_char_p(d0), C.c_void_p).value
            pos, n, warm = 0, C.c_int(0), 0
            while pos < len(d0):                                   # warm-up: the reference loop without the end of stream
                if L.jm_amdintel_need_more_data(h):
                    k = min(L.jm_amdintel_free_buf_len(h), len(d0) - pos)
                    L.jm_amdintel_input_data(base + pos, k, h); pos += k
                n.value = frame_bytes
                if L.jm_amdintel_output_frame(C.cast(out, C.c_void_p), C.byref(n), h) == 0:
                    warm += 1
            L.jm_amddec_set_option(L.jm_amdintel_decoder(h), b"wait_idle", 1)
            timed = d0 * sp
            c_0 = time.perf_counter()
            n1 = L.jm_amdintel_run_pushpull(timed, len(timed), out, frame_bytes, h)
            cdt = time.perf_counter() - c_0
            last_ok = None
            if check_digests[0]:
                last_ok = hashlib.md5(bytes(out)).digest() == check_digests[0][-1]      # the last frame of the stream sits in the caller's buffer
            c0_leg = {"value": round(sp * F / cdt, 1), "unit": "frames/s", "frames": sp * F, "frames_returned": int(n1 + warm), "frames_expected": (sp + 1) * F,
                  "push_bytes": int(L.jm_amdintel_free_buf_len(h)), "bitstream_bytes_per_frame": int(len(d0) / max(F, 1)),
                  "last_frame_equals_checked_pass": last_ok,
                  "vs_single_stream": round(sp * F / cdt / single["value"], 3) if single and single["value"] else None,
                  "decode_errors": int(L.jm_amddec_get_stat(L.jm_amdintel_decoder(h), b"errors")),
                  "note": "BASELINE config 0's shape (test_intel_dec.cpp:64-102): one jm_intel_dec handle, one thread, pushes of free_buf_len, one "
                          "output_frame per loop turn into the caller's buffer (one copy-engine transfer per frame); untimed extra leg"}
            if last_ok is False or n1 + warm != (sp + 1) * F:
                bit_exact = False
                print(f"bench.py: c0_pushpull: {n1 + warm} frames of {(sp + 1) * F}, last frame matches: {last_ok}", file=sys.stderr)
        L.jm_amdintel_deinit(h)

    # ---- untimed: the same S streams with DEVICE-RESIDENT output (jm_amddec_output_frame_device: frames stay in HBM, nothing crosses PCIe on the way
    # out, no frame copy on the CPU) -- the rate the contract calls "inputs and outputs resident in HBM"; `value` above is the PCIe-inclusive one ----
    dev_leg = None
    if world == 1 and not args.parse_only and not args.no_single and not args.device_output:
        hs = []
        for _ in range(S):
            h = jmcodec_amd.jm_nvdec_create_handle()
            L.jm_amddec_set_option(h, b"device_output", 1)
            if jmcodec_amd.jm_nvdec_init(1 if is_hevc else 0, 1, None, 0, h) == 0:
                L.jm_amddec_set_option(h, b"profile", 0 if args.no_profile else 1)   # (the engine's switch follows the handle that set it last: without this the lanes below read 0)
                hs.append(h)
        got_dev = [0] * len(hs)

        def dev_pass(i, passes):
            got_dev[i] = L.jm_amddec_feed_annexb(datas[i], len(datas[i]), passes, None, frame_bytes, hs[i])
            L.jm_amddec_set_option(hs[i], b"wait_idle", 1)
        for passes, timed in ((1, False), (max(1, min(K, 8)), True)):      # (three passes were 0.27 s: a tenth of it pipeline fill and drain)
            dhc0 = host_cpu()
            dl0 = lanes_now(hs[0]) if hs else None
            c0 = time.perf_counter()
            ts = [threading.Thread(target=dev_pass, args=(i, passes)) for i in range(len(hs))]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            if n_dev > 0:
                torch.cuda.synchronize()
            ddt = time.perf_counter() - c0
            if timed:
                dhc1 = host_cpu()
                d_busy = (dhc1["cpu_s"] - dhc0["cpu_s"]) / ddt
                d_quota = dhc1.get("quota_cpus") or os.cpu_count()
                d_lanes = lanes_report(dl0, lanes_now(hs[0]), ddt) if dl0 else {}
                # what bounds it, from measurements instead of by elimination: the host when its CPU allotment is used up; else the
                # device when the busiest lane's stream has kernels on it for >= 85 % of the wall time; else the pipeline in between (batches not formed in time)
                lane_busy = max([v["busy_frac"] for v in d_lanes.values() if isinstance(v, dict) and "busy_frac" in v] or [0.0])
                d_bound = "host_cpu_quota" if d_quota and d_busy >= 0.9 * d_quota else ("gpu" if lane_busy >= 0.85 else ("engine_pipeline" if lane_busy > 0 else "lanes_unmeasured"))
                dev_leg = {"value": round(len(hs) * F * passes / ddt, 1), "unit": "frames/s", "frames": len(hs) * F * passes,
                           "host_cpu": {"cpu_ms_per_frame": round(1e3 * (dhc1["cpu_s"] - dhc0["cpu_s"]) / max(len(hs) * F * passes, 1), 4),
                                        "cpus_busy": round(d_busy, 2), "quota_cpus": d_quota},
                           # (no link in the way here: the host's entropy decode -- >= 90 % of the CPU allotment busy -- or the device)
                           "scaling_bound": d_bound, "busiest_lane_busy_frac": lane_busy, "lanes": d_lanes,
                           "note": "untimed extra leg: the same streams, display frames left in device memory (jm_amddec_output_frame_device, SURVEY 8f f3): "
                                   "no D2H copy, no frame copy on the CPU; host entropy decode and job-list upload still included"}
        for h in hs:
            jmcodec_amd.jm_nvdec_deinit(h)

    if dist is not None:                                      # every rank must have matched
        import torch as _t
        flag = _t.tensor([1.0 if (bit_exact is None or bit_exact) else 0.0, float(frames_checked)], dtype=_t.float64, device=red_dev)
        mn = flag.clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        sm = flag.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        if bit_exact is not None:
            bit_exact = bool(mn[0].item() > 0.5)
        frames_checked = int(sm[1].item())

    if rank != 0:
        ctx["bit_exact_local"] = bit_exact
        return None

    cpu = None
    if world == 1 and not args.no_cpu_baseline and oracle_frames > 0:
        # CPU baseline = the build's own scalar, spec-literal CPU oracle (NOT the reference's libmfx software path, which does not exist in this
        # image: BASELINE.md section 4).  B2: one oracle instance per stream on every core the container owns; B1: one instance on one core.
        cpu = {"value": round(oracle_frames / oracle_dt, 2), "unit": "frames/s", "cores": int(min(host_threads, max(n_oracle - 1, 1))), "kind": "port",
               "sample": f"first IDR period ({len(want[0])} frames) of {n_oracle - first} of the benchmark's own {args.width}x{args.height} streams, one "
               f"oracle instance per stream on "
                         f"{min(host_threads, max(n_oracle - 1, 1))} threads, {oracle_dt:.1f} s wall (build CPU oracle: scalar, spec-literal; not libmfx) -- "
                         f"a label, not a bar",
               "one_core": {"value": round(oracle_one[0] / oracle_one[1], 2), "unit": "frames/s", "cores": 1, "sample": f"{oracle_one[0]} frames of stream "
               f"{stream_ids[0]}, {oracle_one[1]:.1f} s"} if oracle_one else None,
               "host_cpus": os.cpu_count(), "quota_cpus": round(quota_cpus(), 2)}

    value = frames_total / dt_max
    line = {
        "metric": "decoded frames/sec @1080p H.264 + bit-exact YUV" if args.codec == "h264" else f"decoded frames/sec HEVC {args.width}x{args.height} "
        f"(diagnostic, SURVEY 8d C3)",
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
        "config": {"workload": (f"HEVC {tools_desc} + deblocking, IDR every 32, QP 32) " if args.codec == "hevc" else f"H.264 {tools_desc}, IDR every 30, QP "
        f"28, deblock on) ") + f"{args.width}x{args.height}, "
                               f"{S} independent streams per GPU x {F} frames per step, NAL-per-call via jm_nvdec_* API, I420 out",
                   "streams_per_gpu": S, "frames_per_stream_per_step": F, "stream_ids": [stream_ids[0], stream_ids[-1]], "distinct_streams": len(set(datas)),
                   "bitstream_bytes_per_stream": int(sum(len(d) for d in datas) / S), "stream_generation_s": round(gen_s, 1),
                   "host_parse_threads": int(threads), "includes": "host entropy decode + H2D + kernels + packout + D2H into the caller's buffer (one "
                   "copy-engine transfer per frame from device staging, jmcodec_amd/csrc/host_copy.h)"},
        "frames": frames_total,
        "bit_exact": bit_exact, "frames_checked": int(frames_checked), "bit_exact_check": check_note,
        "decode_errors": int(errors),
        "host_ms_per_picture": host_diag,
        "host_cpu": {"cpu_s": round(hc1["cpu_s"] - hc0["cpu_s"], 3), "cpus_busy": round((hc1["cpu_s"] - hc0["cpu_s"]) / dt, 2),
                     "cpu_ms_per_frame": round(1e3 * (hc1["cpu_s"] - hc0["cpu_s"]) / max(frames_local, 1), 4),
                     "quota_cpus": hc1.get("quota_cpus"), "online_cpus": os.cpu_count(),
                     "throttled_ms": round((hc1.get("throttled_usec", 0) - hc0.get("throttled_usec", 0)) / 1e3, 1), "by_thread": by_thread,
                     "calling_threads": {"cpu_ms_per_frame": round(1e3 * sum(feeder_cpu) / max(frames_local, 1), 4),
                     "busiest_thread_share_of_wall": round(max(feeder_cpu) / dt, 3),
                                         "note": "the S threads that call jm_nvdec_decode_frame / jm_nvdec_output_frame (NAL handling, slice headers, DPB, "
                                         "the frame copy); 1.0 = a handle's own thread is what bounds it"},
                     "cpu_needed_for_8_gpus": round(8 * (hc1["cpu_s"] - hc0["cpu_s"]) / dt, 1),
                     "note": "rank 0, timed region; when cpus_busy sits at quota_cpus the host half (entropy decode) bounds the rate; "
                             "cpu_needed_for_8_gpus = 8 x cpus_busy is what an 8-rank run of this rate would need from the node"},
        "roofline": {"bound": "hbm", "kernel": "k_" + dominant, "achieved": round(achieved, 3), "peak": peak, "unit": "GB/s",
                     "frac": round(achieved / peak, 6), "traffic": traffic, "traffic_raw": traffic_raw,
                     "traffic_file": traffic_file,
                     "traffic_note": "HBM bytes per launch from PMC counters (traffic_file: separate --pmc FETCH_SIZE / WRITE_SIZE passes of this codec / "
                     "tool set / size, per picture) x "
                                     "pictures per launch of THIS run; traffic = 2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction for wide "
                                     "coalesced reads, an "
                                     "upper estimate for these access shapes), traffic_raw = FETCH_SIZE + WRITE_SIZE as counted",
                     "alg_bytes_per_launch": int(alg[dominant]), "avg_launch_us": round(avg_s[dominant] * 1e6, 2),
                     "launches": int(tot_n[dominant]), "pictures_per_launch": round(tot_pics[dominant] / max(tot_n[dominant], 1), 2)},
        "engine": {"batches": int(batches), "pictures_per_batch": round(batch_pics / max(batches, 1), 2), "engine_thread_ms": eng_thread_ms,
        "formation": form_stat, "lanes": lanes_timed, "direct_output": direct_stat,
                   "chain_batches_whole_run": int(chain_stat[0]), "chain_pictures_whole_run": int(chain_stat[1]), "device_wait_errors": int(chain_stat[2]),
                   "chain_recoveries_whole_run": int(chain_stat[3]), "gpu_shared_with_another_process": bool(chain_stat[4]),
                   "chain_batches": int(chain_stat[0]), "chain_i_batches": int(chain_stat[5]), "chain_recoveries": int(chain_stat[3]),
                   # chain launches whose waits saw the wall clock jump by more than 5 ms between two looks: their waves were not run meanwhile (the timers
                   # leave such gaps out: chain_common.h WaitClock); the longest jump any launch of this process saw
                   "chain_launches_with_clock_gaps": int(chain_stat[6]), "longest_clock_gap_us_whole_process": int(gap_max_us)},
        "pcie_out": None if args.device_output else {"bound": "pcie", "achieved": round(value / world * frame_bytes / 1e9, 2), "peak": 54.0, "unit": "GB/s",
        "frac": round(value / world * frame_bytes / 1e9 / 54.0, 4),
                     "note": "what bounds the rate WITH host output: every frame crosses the link once (k_packout -> device staging -> copy engine -> "
                     "caller's buffer); peak = device->host "
                             "rate measured on this platform with three SDMA engines in turn, four copies in flight (tools/sdma_probe.cpp, "
                             "profiles/r02_sdma_probe.txt: 54.0 GB/s = 17.4 k frames/s of 1080p; two engines 52.5, one 47); "
                             "achieved = frames/s x frame bytes per GPU"},
        "kernels": {("k_" + k): {"launches": int(tot_n[k]), "avg_us": round(avg_s[k] * 1e6, 2), "pictures_per_launch": round(tot_pics[k] / max(tot_n[k], 1), 2),
                                 "alg_GBps": round(alg[k] / avg_s[k] / 1e9, 2) if avg_s[k] > 0 else None} for k in names},
        "roofline_frame": {"alg_bytes_per_frame": int(A), "job_bytes_per_frame": int(J),
                           "end_to_end_GBps": round(value / world * A / 1e9, 3), "end_to_end_frac": round(value / world * A / 1e9 / peak, 6),
                           "kernel_time_GBps": round(A / kernel_s_per_frame / 1e9, 3) if kernel_s_per_frame > 0 else None},
    }
    # host memory of this rank (pinned job and output buffers of S handles included): what N ranks need from the node's memory allotment
    try:
        import resource
        mem = {"peak_rss_mb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)}
        for name, key in (("memory.current", "cgroup_current_mb"), ("memory.max", "cgroup_max_mb")):
            try:
                v = open("/sys/fs/cgroup/" + name).read().strip()
                mem[key] = None if v == "max" else round(int(v) / 1048576.0, 1)
            except (OSError, ValueError):
                pass
        mem["peak_rss_timed_region_mb"] = peak_rss_timed_mb
        mem["needed_for_8_gpus_mb"] = round(8 * peak_rss_timed_mb, 0)
        mem["job_slots_mb"] = job_slot_mb
        mem["job_slots_grown"] = job_regrown
        mem["note"] = ("peak_rss_mb includes the untimed check pass and the CPU oracle; peak_rss_timed_region_mb is the high-water mark when the timed region "
                       "ends (Python / torch runtime, the bitstreams, the handles) and what needed_for_8_gpus_mb multiplies; job_slots_mb = page-locked job "
                       "buffers of this "
                       "rank's handles (ordinary-picture size; three worst-case buffers per handle are lent to I pictures)")
        line["host_memory"] = mem
    except Exception:
        pass
    # what bounds this rank's rate, from three measured utilisations instead of by elimination: the PCIe link (frames leave at this fraction of the measured
    # device->host rate), the host CPU allotment (share of the quota the process keeps busy), the device (share of the wall time the busiest engine lane's
    # stream has kernels on it).  The highest one names the bound when it is >= 0.85; when nothing is that busy the pipeline in between is (batches formed late,
    # uploads, callers): "engine_pipeline".
    q_cpus = hc1.get("quota_cpus") or os.cpu_count()
    util = {"pcie": line["pcie_out"]["frac"] if line["pcie_out"] else None,
            "host_cpu_quota": round(line["host_cpu"]["cpus_busy"] / (q_cpus / max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))), 3) if q_cpus else None,
            "gpu": max([v["busy_frac"] for v in lanes_timed.values() if isinstance(v, dict) and "busy_frac" in v] or [None], key=lambda x: -1 if x is None else x)}
    known = {k: v for k, v in util.items() if v is not None}
    top = max(known, key=known.get) if known else None
    bound = top if top is not None and known[top] >= 0.85 else ("engine_pipeline" if known else None)
    line["bound_utilisation"] = util
    line["scaling_bound"] = bound
    line["numa_node"] = numa_node
    if cpu is not None:
        line["cpu_baseline"] = cpu
    if single is not None:
        line["single_stream"] = single
    if c0_leg is not None:
        line["c0_pushpull"] = c0_leg
    if dev_leg is not None:
        line["device_resident_output"] = dev_leg
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    q = hc1.get("quota_cpus") or os.cpu_count()
    if world > 1 and q and line["host_cpu"]["cpus_busy"] * local_world > 0.95 * q:
        line["host_cpu"]["oversubscribed"] = True
        print(f"bench.py: WARNING: {local_world} ranks x {line['host_cpu']['cpus_busy']} busy CPUs meet the node's {q} CPUs -- this run is bound by the host "
              f"(entropy decode), not by the GPUs; the scaling figure measures the CPU allotment", file=sys.stderr)
    return line


if __name__ == "__main__":
    main()
