"""Developer script: idle gaps on the ordinary-picture lane, from a rocprofv3 --kernel-trace CSV.
For every batched k_deblock_lds launch (grid y = pictures) prints what happened until the next k_recon_inter of the same queue."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id'],
    int(r['Grid_Size_Y']) if 'Grid_Size_Y' in r else 0) for r in rows)
byq = collections.defaultdict(list)
for e in ev:
    byq[e[3]].append(e)
for q, lst in byq.items():
    names = collections.Counter(n.split('(')[0].split('::')[-1][:24] for _, _, n, _, _ in lst)
    print("queue", q, dict(names))
    gaps, durs = [], collections.defaultdict(list)
    for i in range(len(lst) - 1):
        s, e, n, _, gy = lst[i]
        durs[n.split('(')[0].split('::')[-1][:24]].append((e - s) / 1e3)
        gaps.append((lst[i + 1][0] - e) / 1e3)
    if gaps:
        gaps_sorted = sorted(gaps)
        print("   gaps between consecutive kernels (us): median %.1f p90 %.1f max %.1f sum %.1f ms of %.1f ms" % (
            gaps_sorted[len(gaps) // 2], gaps_sorted[int(len(gaps) * .9)], gaps_sorted[-1], sum(g for g in gaps if g > 0) / 1e3,
            (lst[-1][1] - lst[0][0]) / 1e6))
        for k, v in durs.items():
            v.sort(); print("   %-26s n=%d median %.1f us" % (k, len(v), v[len(v) // 2]))

# duration of the batched kernels by batch size (grid y)
by = collections.defaultdict(list)
for s0, e0, n, q, gy in ev:
    nm = n.split('(')[0].split('::')[-1][:22]
    if nm.startswith('k_recon_inter') or nm.startswith('k_deblock_lds') or nm.startswith('k_intra_lds'):
        by[(nm, 'full' if gy >= 30 else ('mid' if gy >= 12 else 'small'))].append((e0 - s0) / 1e3)
for k in sorted(by):
    v = sorted(by[k]); print("%-24s %-5s n=%4d median %8.1f us" % (k[0], k[1], len(v), v[len(v) // 2]))
