"""Developer script: sum rocprofv3 --pmc counters per kernel name from *counter_collection.csv."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].split('::')[-1][:22]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k, {c: round(v / 1e6, 2) for c, v in d.items()}, "launches", max(cnt[(k, c)] for c in d))
