"""Developer script: print a window of the kernel timeline (all queues) around full batches."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ts = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-16:], r['Queue_Id'], r['Grid_Size_X'],
    r['Grid_Size_Y']) for r in rows]
ts.sort()
t0 = ts[0][0]
idx = [i for i, t in enumerate(ts) if 'packout' in t[2] and t[1] - t[0] > 800000]
i = idx[len(idx) // 2] if idx else len(ts) // 2
for s, e, n, q, gx, gy in ts[max(0, i - 8):i + 22]:
    print("%10.3f %10.3f %8.1f %-16s q%s grid %sx%s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, n, q, gx, gy))
