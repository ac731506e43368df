# round 4, final call: GPU suite, smoke(), two random sweeps, the profile set (scratch/gpu_prof_r04.sh: default line with its legs, per-configuration
# kernel stats + PMC traffic), one / eight stream lines, four more C2 runs with chains forced on
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/f4
timeout 1500 python -m pytest tests -m gpu -q -rs > gpurun_out/f4/gputests.log 2>&1; tail -4 gpurun_out/f4/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python tools/gpu_sweep.py 200 61 > gpurun_out/f4/sweep_a.log 2>&1; tail -2 gpurun_out/f4/sweep_a.log
timeout 1500 python tools/gpu_sweep.py 200 62 > gpurun_out/f4/sweep_b.log 2>&1; tail -2 gpurun_out/f4/sweep_b.log
timeout 900 python tools/gpu_sweep.py 30 63 big > gpurun_out/f4/sweep_big.log 2>&1; tail -2 gpurun_out/f4/sweep_big.log
bash scratch/gpu_prof_r04.sh r04 > gpurun_out/f4/prof.log 2>&1; tail -30 gpurun_out/f4/prof.log | cut -c1-260
for s in 1 8; do timeout 300 python bench.py --streams $s --no-extra --no-cpu-baseline --no-single > gpurun_out/f4/r04_bench_s$s.json 2>/dev/null; done
for t in high high_b; do timeout 300 python bench.py --tools $t --no-extra --no-cpu-baseline --no-single > gpurun_out/f4/r04_bench_$t.json 2>/dev/null; done
C2="--tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-extra --no-cpu-baseline --no-single"
for i in 7 8 9 10; do JM_AMD_DEC_CHAIN_STREAMS=64 JM_AMD_DEC_VERBOSE=1 timeout 300 python bench.py $C2 > gpurun_out/f4/c2_chain_$i.json 2> gpurun_out/f4/c2_chain_$i.err; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/f4/*.json')) + sorted(glob.glob('gpurun_out/p4/r04_*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(f, 'unreadable'); continue
    if "value" not in d: continue
    print(os.path.basename(f), d['value'], d.get('scaling_bound'), d['host_cpu']['cpu_ms_per_frame'], d['host_cpu']['cpus_busy'], d['bit_exact'], "recov", d["engine"]["chain_recoveries_whole_run"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
