"""Developer script: symbolises a JM_HOST_BENCH_PROF sample file (tools/host_bench.cpp) -- per function (outermost non-inlined frame), per
inlined function and per source line.  usage: python scratch/hostprof.py samples.txt [binary]"""
import collections, subprocess, sys
path = sys.argv[1]; binary = sys.argv[2] if len(sys.argv) > 2 else "tools/_build/host_bench"
addrs = [hex(int(l, 16) + 0x200000) for l in open(path) if not l.startswith("other")]
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "--obj=" + binary, "-f", "-C", "-i"], input="\n".join(addrs), capture_output=True,
    text=True).stdout
outer, inner, lines = collections.Counter(), collections.Counter(), collections.Counter()
n = 0
for rec in out.split("\n\n"):
    l = [x for x in rec.split("\n") if x]
    if len(l) < 2: continue
    n += 1
    f = lambda t: t.replace("(anonymous namespace)::",
        "").split("(")[0][-50:]; outer[f(l[-2])] += 1; inner[f(l[0])] += 1; lines[l[1].split("/")[-1].rsplit(":", 1)[0]] += 1
print(n, "samples")
for title, c, k in (("function", outer, 14), ("innermost inlined", inner, 24), ("line", lines, 60)):
    print("--", title)
    for name, v in c.most_common(k): print("%6.2f%%  %s" % (100.0 * v / n, name))
