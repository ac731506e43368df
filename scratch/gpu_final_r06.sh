# round 6, final call: GPU suite, smoke(), random sweeps (2 x 400 + 60 big), the profile set (scratch/gpu_prof_r06.sh: default line with its legs,
# per-configuration kernel stats + PMC traffic), SQ counters of the final tree, one / eight stream lines, High / High + B lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/f6
timeout 1500 python -m pytest tests -m gpu -q -rs > gpurun_out/f6/gputests.log 2>&1; tail -4 gpurun_out/f6/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 2400 python tools/gpu_sweep.py 400 101 > gpurun_out/f6/sweep_a.log 2>&1; tail -2 gpurun_out/f6/sweep_a.log
timeout 2400 python tools/gpu_sweep.py 400 102 > gpurun_out/f6/sweep_b.log 2>&1; tail -2 gpurun_out/f6/sweep_b.log
timeout 1500 python tools/gpu_sweep.py 60 103 big > gpurun_out/f6/sweep_big.log 2>&1; tail -2 gpurun_out/f6/sweep_big.log
bash scratch/gpu_prof_r06.sh r06 > gpurun_out/f6/prof.log 2>&1; tail -30 gpurun_out/f6/prof.log | cut -c1-260
bash scratch/gpu_sq_r05.sh final jmcodec_amd/lib/libjm_amd_dec.so c1 c2 c3 > gpurun_out/f6/sq.log 2>&1; grep -c "^k_" gpurun_out/f6/sq.log
for s in 1 8; do timeout 300 python bench.py --streams $s --no-extra --no-cpu-baseline --no-single > gpurun_out/f6/r06_bench_s$s.json 2>/dev/null; done
for t in high high_b; do timeout 300 python bench.py --tools $t --no-extra --no-cpu-baseline --no-single > gpurun_out/f6/r06_bench_$t.json 2>/dev/null; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/f6/*.json')) + sorted(glob.glob('gpurun_out/p6/r06_*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(f, 'unreadable'); continue
    if "value" not in d: continue
    print(os.path.basename(f), d['value'], d.get('scaling_bound'), d['host_cpu']['cpu_ms_per_frame'], d['host_cpu']['cpus_busy'], d['bit_exact'], "recov", d["engine"]["chain_recoveries_whole_run"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
