#!/bin/bash
# round 6: (a) SQ counters of the C3 kernels after the packed-arithmetic rewrite, (b) C3 4K kernel stats (16 streams, device-resident),
# (c) host-output headline with and without intra pictures ahead of their turn, interleaved
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f gpurun_out/sq5; O=gpurun_out/f
bash scratch/gpu_sq_r05.sh r06 jmcodec_amd/lib/libjm_amd_dec.so c3 > $O/sq_c3.log 2>&1
cp gpurun_out/sq5/r06_c3.json $O/r06_sq_counters_c3.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k -- python3 bench.py --codec hevc --width 3840 --height 2160 --streams 16 --frames 16 --steps 3 --device-output --no-cpu-baseline --no-single --no-extra > $O/r06_hevc_3840x2160_device_output_under_rocprof.json 2>/dev/null
find gpurun_out/prof_k -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r06_hevc_3840x2160_kernel_stats.csv; rm -rf gpurun_out/prof_k
for i in 1 2 3; do
  JM_AMD_DEC_CROSS_LANE=0 JM_AMD_DEC_EARLY_INTRA=0 python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_00_$i.json 2> $O/host_00_$i.err
  python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_11_$i.json 2> $O/host_11_$i.err
  JM_AMD_DEC_CROSS_LANE=1 JM_AMD_DEC_EARLY_INTRA=0 python bench.py --no-extra --no-cpu-baseline --no-single > $O/host_10_$i.json 2> $O/host_10_$i.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/f/host_*.json')):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'NO LINE'); continue
    e = d["engine"]; ln = e.get("lanes", {}); o = ln.get("ordinary", {}); it = ln.get("intra", {})
    print(os.path.basename(f), d["value"], d["bit_exact"], d["scaling_bound"], "cpus", d["host_cpu"]["cpus_busy"], "ord", o.get("pictures_per_batch"), o.get("busy_frac"), "intra", it.get("batches"), it.get("pictures_per_batch"), "early", ln.get("intra_pictures_launched_ahead_of_their_turn"))
PY
head -14 $O/r06_hevc_3840x2160_kernel_stats.csv | cut -c1-160
python - <<'PY'
import json
d=json.load(open('gpurun_out/f/r06_sq_counters_c3.json'))
for tag, v in d.items():
    if isinstance(v, dict) and 'kernels' in v:
        for k, c in v['kernels'].items(): print(tag, k, int(c.get('SQ_INSTS_VALU',0)), int(c.get('SQ_INSTS_SALU',0)), int(c.get('SQ_WAVES',0)), c.get('avg_us_under_pmc'), c.get('wait_share'))
PY
