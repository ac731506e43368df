cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hevc -- python3 bench.py --codec hevc --streams 8 --frames 16 --width 3840 --height 2160 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_hevc.json 2> gpurun_out/prof_hevc.err
find gpurun_out/prof_hevc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/hevc_kernel_stats.csv
find gpurun_out/prof_hevc -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 -c "
import csv,sys
rows=list(csv.DictReader(open('{}')))
import collections
d=collections.defaultdict(list)
for r in rows: d[r['Kernel_Name'][:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for k,v in d.items():
    v2=sorted(v); print(k, len(v), 'min %.0f med %.0f p90 %.0f max %.0f us' % (v2[0], v2[len(v2)//2], v2[int(len(v2)*0.9)], v2[-1]))
"
rm -rf gpurun_out/prof_hevc
cat gpurun_out/hevc_kernel_stats.csv | head -12
