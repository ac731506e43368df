"""Developer script: kernel-level concurrency histogram + per-kernel busy time from a rocprofv3 kernel trace CSV."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t_first = min(int(r["Start_Timestamp"]) for r in rows if "k_recon_inter" in r["Kernel_Name"])
t_last = max(int(r["End_Timestamp"]) for r in rows)
t_first = t_last - (t_last - t_first) * 4 // 10                          # steady state: the last 40 % of the decode phase
rows = [r for r in rows if int(r["Start_Timestamp"]) >= t_first]
qbusy = collections.Counter()
ev = []
per = collections.defaultdict(lambda: [0, 0])
qs = set()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1)); qs.add(r["Queue_Id"])
    n = r["Kernel_Name"].split("(")[0]; per[n][0] += e - s; per[n][1] += 1; qbusy[r["Queue_Id"]] += e - s
ev.sort()
cur, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("kernels", len(rows), "queues", len(qs), "span ms %.1f" % (tot / 1e6))
print("concurrency histogram (% of span):", {k: round(100 * v / tot, 1) for k, v in sorted(hist.items())})
print("mean concurrency %.2f" % (sum(k * v for k, v in hist.items()) / tot))
for n, (t, c) in sorted(per.items(), key=lambda x: -x[1][0]):
    print("  %-40s calls %6d total ms %9.1f avg us %9.1f" % (n[-40:], c, t / 1e6, t / c / 1e3))
print("per-queue busy % of span:", sorted(round(100 * v / tot, 1) for v in qbusy.values()))
# gaps between consecutive kernels of the same queue, by transition
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1]))
gaps = collections.defaultdict(lambda: [0, 0, 0]); gl = collections.defaultdict(list)
for q, ks in byq.items():
    ks.sort()
    for a, b in zip(ks, ks[1:]):
        g = b[0] - a[1]
        key = a[2] + " -> " + b[2]
        gaps[key][0] += g; gaps[key][1] += 1; gaps[key][2] = max(gaps[key][2], g); gl[key].append(g)
print("gaps within a queue (avg us, max us, count):")
for k, (t, c, m) in sorted(gaps.items(), key=lambda x: -x[1][0])[:12]:
    v = sorted(gl[k]); print("  %-40s avg %8.1f p50 %8.1f p90 %8.1f max %9.1f n %d" % (k, t / c / 1e3, v[len(v) // 2] / 1e3, v[len(v) * 9 // 10] / 1e3,
        m / 1e3, c))
