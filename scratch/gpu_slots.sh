# job slot allowance for Main / High streams: grow events, kernel time of the parse workers and rate, lines written as profiles/r03_* files
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sl
for cfg in high high_b paff paff_b; do python bench.py --tools $cfg --no-cpu-baseline --no-single > gpurun_out/sl/r3_$cfg.json 2>/dev/null; done
python bench.py --tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3 --no-cpu-baseline --no-single > gpurun_out/sl/r3_c2.json 2>/dev/null
python bench.py --no-cpu-baseline > gpurun_out/sl/r3_default.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/sl/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(f, 'unreadable'); continue
    hc = d['host_cpu']; hm = d.get('host_memory', {})
    print(os.path.basename(f), d['value'], d.get('scaling_bound'), hc['cpu_ms_per_frame'], hc['cpus_busy'], hc['by_thread'].get('jm-parse'), 'grown', hm.get('job_slots_grown'), 'slots_mb', hm.get('job_slots_mb'), 'rss', hm.get('peak_rss_timed_region_mb'), d['bit_exact'])
PY
