# per-dispatch durations of the HEVC kernels for ONE stream (kernel trace), to see which pictures cost what
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ht; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ht; rocprofv3 --kernel-trace -d /tmp/ht -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --codec hevc --streams 1 --frames 32 --steps 2 --warmup 1 --no-cpu-baseline --no-single --device-output > $GRAFT_REPO_ROOT/gpurun_out/ht/line.json 2> $GRAFT_REPO_ROOT/gpurun_out/ht/err.txt
f=$(find /tmp/ht -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].split('(')[0].split('::')[-1]
    by[n].append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
for n, v in by.items():
    d = sorted(x[1] for x in v)
    print(n, len(d), "sum_ms %.1f" % (sum(d) / 1e3), "min %.0f med %.0f p90 %.0f max %.0f" % (d[0], d[len(d) // 2], d[int(len(d) * .9)], d[-1]))
v = sorted(by.get('k_hevc_intra', []))
print("k_hevc_intra in order:", [round(x[1]) for x in v[-70:]])
PY
