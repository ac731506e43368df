#!/bin/bash
# round 5: does the host parser slow down when many parse threads run side by side (memory traffic / SMT)?  tools/host_bench, N copies at once.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/hostpar; mkdir -p $O
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "model name\|thread\|core\|socket\|L2\|L3\|numa"; } > $O/cpu.txt 2>&1
python - <<'PY'
import sys
sys.path.insert(0, 'tools')
import streams
open('/tmp/c2_4k.h264', 'wb').write(streams.generate(**streams.config_c2(frames=24)))
open('/tmp/c1.h264', 'wb').write(streams.generate(**streams.config_c1(frames=60)))
open('/tmp/c3_4k.hevc', 'wb').write(streams.generate_hevc(**streams.config_c3(frames=16)))
PY
make -C tools host_bench >/dev/null 2>&1 || true
ls -la tools/_build/host_bench /tmp/c2_4k.h264 /tmp/c1.h264 /tmp/c3_4k.hevc >> $O/cpu.txt 2>&1
run() { # name file codec passes
  for n in 1 4 8 16 32; do
    for i in $(seq $n); do tools/_build/host_bench $2 $4 $3 2>&1 | tail -1 | sed "s/^/$1 n=$n /" >> $O/$1.txt & done; wait
  done
}
run c2_4k /tmp/c2_4k.h264 0 6
run c1 /tmp/c1.h264 0 10
run c3_4k /tmp/c3_4k.hevc 1 6
for f in c2_4k c1 c3_4k; do python - $O/$f.txt <<'PY'
import sys, re, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.search(r'n=(\d+) .* ([\d.]+) ms per picture', l)
    if m: d[int(m.group(1))].append(float(m.group(2)))
for n in sorted(d): print(sys.argv[1].split('/')[-1], 'copies', n, 'ms per picture: mean %.3f min %.3f max %.3f' % (sum(d[n]) / len(d[n]), min(d[n]), max(d[n])))
PY
done | tee $O/summary.txt
cat $O/cpu.txt
echo finished
