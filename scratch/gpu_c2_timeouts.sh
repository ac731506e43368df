# C2 several times with the engine's verbose messages: which wait of a chain launch gives up (the codes of CHAIN_ERR_*)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/to
C2="--tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-cpu-baseline --no-single"
for rep in 1 2 3 4 5 6; do
  JM_AMD_DEC_VERBOSE=1 python bench.py $C2 --device-output > gpurun_out/to/c2_$rep.json 2> gpurun_out/to/c2_$rep.err
  grep -c "gave up" gpurun_out/to/c2_$rep.err; grep "gave up" gpurun_out/to/c2_$rep.err | head -3 | cut -c1-300
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/to/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(f, 'unreadable'); continue
    e = d['engine']; print(os.path.basename(f), d['value'], e['chain_recoveries_whole_run'], e['device_wait_errors'], d.get('decode_errors'), d['kernels']['k_chain'])
PY
