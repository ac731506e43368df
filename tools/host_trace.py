"""Developer script: per-picture host pipeline timeline from JM_AMD_DEC_TRACE dumps."""
import csv, glob, sys
files = sorted(glob.glob(sys.argv[1] + "*.csv"))
f = files[0]
rows = [dict((k, int(v)) for k, v in r.items()) for r in csv.DictReader(open(f))]
t0 = rows[0]["dispatch"]
print(f, len(rows), "pictures; times in ms relative to first dispatch")
prev = None
for r in rows[60:150]:
    d = (r["dispatch"] - t0) / 1e6; p = (r["parsed"] - t0) / 1e6; s0 = (r["submit0"] - t0) / 1e6; s1 = (r["submit1"] - t0) / 1e6
    gap = (s0 - prev) if prev is not None else 0
    print("seq %4d %s dispatch %9.2f parse %6.2f ms  parsed->submit %7.2f  submit dur %5.2f  since prev submit %7.2f" % (r["seq"], "I" if r["is_i"] else "P",
        d, p - d, s0 - p, s1 - s0, gap))
    prev = s0
