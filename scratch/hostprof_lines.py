"""Developer script: samples of a JM_HOST_BENCH_PROF file per source line of ONE function, from objdump's line markers (llvm-symbolizer files much of an
optimised function under line 0).  usage: python scratch/hostprof_lines.py samples.txt function-substring [binary]"""
import collections, subprocess, sys
path, fn = sys.argv[1], sys.argv[2]; binary = sys.argv[3] if len(sys.argv) > 3 else "tools/_build/host_bench"
addrs = [int(l, 16) + 0x200000 for l in open(path) if not l.startswith("other")]
c = collections.Counter(addrs)
lines = [l.split(None, 2) for l in subprocess.run(["nm", "-C", "--defined-only", "-n", binary], capture_output=True, text=True).stdout.splitlines()]
lo = hi = None
for i, l in enumerate(lines):
    if len(l) == 3 and fn in l[2] and l[1] in "tT" and lo is None:
        lo = int(l[0], 16); hi = next(int(m[0], 16) for m in lines[i + 1:] if len(m) == 3 and int(m[0], 16) > lo)
dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", "-l", "--start-address=" + hex(lo), "--stop-address=" + hex(hi),
    binary],
    capture_output=True, text=True).stdout
per = collections.Counter(); cur = "?"
for l in dis.splitlines():
    if l.startswith("; /"): cur = l.split("/")[-1].split(" ")[0]; continue
    p = l.strip().split(":", 1)
    try: a = int(p[0], 16)
    except ValueError: continue
    per[cur] += c.get(a, 0)
tot = sum(per.values())
print(fn, tot, "samples of", len(addrs), "= %.1f%%" % (100.0 * tot / len(addrs)))
for k, v in sorted(per.items(), key=lambda kv: (kv[0].split(":")[0], int(kv[0].split(":")[1]) if ":" in kv[0] and kv[0].split(":")[1].isdigit() else 0)):
    if v * 500 > tot: print("%6.2f%%  %s" % (100.0 * v / len(addrs), k))
