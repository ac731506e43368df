"""Multi-GPU path: streams shard across ranks with no data-path collective; only the timing / frame
count reduction of bench.py uses torch.distributed.  Exercised here with 2 gloo ranks on the CPU (parse_only)."""
import os
import socket

import pytest

from jmcodec_amd import shard


def test_assignment_is_a_partition():
    for total in (1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            parts = [shard.streams_of_rank(total, r, world) for r in range(world)]
            flat = sorted(s for p in parts for s in p)
            assert flat == list(range(total))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard.streams_of_rank(64, 3, 8) == [3, 11, 19, 27, 35, 43, 51, 59]     # stream i -> GPU i mod n


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jmcodec_amd import api
    from tools import streams
    mine = shard.streams_of_rank(6, rank, world)
    frames = 0
    for sid in mine:
        data = streams.generate(width=64, height=48, frames=3 + sid % 2, gop=4, seed=100 + sid)
        with api.JmAmdDec(0, 1, options={"parse_only": 1}) as d:
            frames += d.decode_stream(data, keep=False)
    total, tmax = shard.reduce_result(dist, frames, 1.0 + rank, device="cpu")
    q.put((rank, frames, total, tmax))
    dist.destroy_process_group()


def test_two_rank_reduction_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want_total = sum(3 + sid % 2 for sid in range(6))
    assert [r[2] for r in res] == [want_total, want_total]
    assert [r[3] for r in res] == [2.0, 2.0]                     # max over ranks
    assert res[0][1] + res[1][1] == want_total


def test_bench_two_ranks_gloo_parse_only():
    """bench.py itself under the driver's multi-rank launch line (torch.distributed.run, 2 ranks, gloo, no GPU: --parse-only): the S*world distinct
    streams shard as stream i -> rank i mod world, rank 0 prints ONE JSON line whose frame count is the sum over ranks, the bit-exact flag is reduced."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--parse-only", "--streams", "3", "--frames", "6", "--width",
           "176", "--height", "144"]
    env = dict(os.environ, JM_BENCH_CACHE=os.environ.get("TMPDIR", "/tmp"))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["frames"] == 2 * 3 * 6 * 2                       # ranks x streams x frames x steps
    assert d["config"]["stream_ids"] == [0, 4]                # rank 0 of 2: streams 0, 2, 4 of the job's 6
    assert d["bit_exact"] is None and "cpu_baseline" not in d  # parse-only: no pixels to check; the CPU baseline is an N=1 leg


def test_bench_extra_legs_parse_only(tmp_path):
    """the default bench invocation appends untimed legs for the other BASELINE configurations -- c4_slice (8 x 1080p Baseline), c2_4k
    (16 x 4K High I B B P), c3_4k (16 x 4K HEVC) -- each a compact object with its own bit-exact flag and roofline block.  Here: the same code path at a toy
    size without a GPU (--parse-only, JM_BENCH_TEST_LEGS): one JSON line, the headline keys unchanged, the three legs present with the promised fields."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--parse-only", "--streams", "2", "--frames", "8", "--width", "176",
           "--height", "144"]
    env = dict(os.environ, JM_BENCH_CACHE=str(tmp_path), JM_BENCH_TEST_LEGS="176x144")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["frames"] == 2 * 8 and d["n_gpus"] == 1
    want = {"c4_slice": ("H.264 Baseline", 60), "c2_4k": ("H.264 High, I B B P", 24), "c3_4k": ("HEVC Main", 16)}
    for key, (tools, frames) in want.items():
        leg = d[key]
        assert "error" not in leg, leg
        assert leg["workload"].startswith(tools) and f"x {frames} frames per step" in leg["workload"]
        assert leg["value"] > 0 and leg["decode_errors"] == 0
        for k in ("bit_exact", "frames_checked", "scaling_bound", "roofline", "host_cpu", "engine", "kernels"):
            assert k in leg
        for k in ("kernel", "frac", "traffic", "alg_bytes_per_launch", "avg_launch_us"):
            assert k in leg["roofline"]
        assert "cpu_ms_per_frame" in leg["host_cpu"]


def test_bench_eight_ranks_gloo_parse_only():
    """BASELINE config 5 at its real SHAPE on a host without a GPU: bench.py under the driver's launch line with 8 ranks (gloo, --parse-only), 8 streams per
    rank -- 64 distinct streams, stream i -> rank i mod 8 (the reference is single-device, nv_dec/nv_dec.cpp:209), LOCAL_WORLD_SIZE 8 divides the
    parse pool, rank 0 prints ONE JSON line whose frame count is the sum over the eight ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--parse-only", "--streams", "8", "--frames", "4", "--width",
           "64", "--height", "48"]
    env = dict(os.environ, JM_BENCH_CACHE=os.environ.get("TMPDIR", "/tmp"), OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert d["frames"] == 8 * 8 * 4                            # ranks x streams x frames x steps
    assert d["config"]["stream_ids"] == [0, 56]                # rank 0 of 8: streams 0, 8, .., 56 of the job's 64
    assert d["config"]["streams_per_gpu"] == 8
    # the parse pool of a rank is its share of the node's CPUs: quota x 1.25 / LOCAL_WORLD_SIZE workers, never fewer than four (bench.py setup_process)
    import bench
    assert d["config"]["host_parse_threads"] == max(4, min(64, int(bench.quota_cpus() * 1.25 + 0.5) // 8))
