# parse pool bound to the GPU's NUMA node (default) against unbound (JM_AMD_DEC_FAKE_NUMA names no node for device 0), one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/na
{ lscpu | grep -i -E "numa|socket|core|thread|model name"; for n in /sys/devices/system/node/node*; do echo "$n $(cat $n/cpulist)"; done; cat /sys/bus/pci/devices/*/numa_node 2>/dev/null | sort | uniq -c; } > gpurun_out/na/topology.txt 2>&1
for rep in 1 2; do
  for cfg in high high_b; do
    python bench.py --tools $cfg --steps 3 --no-cpu-baseline --no-single > gpurun_out/na/bound_${cfg}_$rep.json 2>/dev/null
    JM_AMD_DEC_FAKE_NUMA="99:0" python bench.py --tools $cfg --steps 3 --no-cpu-baseline --no-single > gpurun_out/na/unbound_${cfg}_$rep.json 2>/dev/null
  done
done
python bench.py --codec hevc --streams 16 --frames 32 --steps 3 --no-cpu-baseline --no-single > gpurun_out/na/bound_hevc.json 2>/dev/null
JM_AMD_DEC_FAKE_NUMA="99:0" python bench.py --codec hevc --streams 16 --frames 32 --steps 3 --no-cpu-baseline --no-single > gpurun_out/na/unbound_hevc.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/na/*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(f, 'unreadable'); continue
    hc = d['host_cpu']; print(os.path.basename(f), d['value'], d.get('numa_node'), hc['cpu_ms_per_frame'], hc['cpus_busy'], hc.get('throttled_ms'), hc['by_thread'].get('jm-parse'))
PY
head -30 gpurun_out/na/topology.txt
