# the chaining decision weighs a stream by its picture size: C2 several times (no chain launches expected, no give-ups), the chain tests, one 4K stream, the default line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ld
C2="--tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-cpu-baseline --no-single"
for rep in 1 2 3 4; do JM_AMD_DEC_VERBOSE=1 python bench.py $C2 > gpurun_out/ld/c2_$rep.json 2> gpurun_out/ld/c2_$rep.err; grep -c "gave up" gpurun_out/ld/c2_$rep.err; done
python bench.py --tools high_b --width 3840 --height 2160 --streams 2 --frames 24 --steps 3 --no-cpu-baseline --no-single > gpurun_out/ld/c2_2streams.json 2>/dev/null
python bench.py --no-cpu-baseline > gpurun_out/ld/bench.json 2>/dev/null
timeout 1200 python -m pytest tests -m gpu -q -x -k "chain or recover or stream" > gpurun_out/ld/gputests.log 2>&1; tail -2 gpurun_out/ld/gputests.log
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/ld/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(f, 'unreadable'); continue
    e = d['engine']; print(os.path.basename(f), d['value'], d['scaling_bound'], d['host_cpu']['cpus_busy'], d['bit_exact'], 'chains', e['chain_batches_whole_run'], 'recov', e['chain_recoveries_whole_run'], 'errs', d.get('decode_errors'),
        {k: (v['launches'], v['avg_us'], v['pictures_per_launch']) for k, v in d['kernels'].items() if v['launches']})
PY
