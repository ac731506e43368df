# round 4: how many parse workers for a host-bound load on a 16-CPU quota?  (default: 1.25 x the quota = 20)  C3 1080p / 4K HEVC and C2 4K High, 16 streams
cd $GRAFT_REPO_ROOT; P=gpurun_out/th; mkdir -p $P
python bench.py --codec hevc --width 1920 --height 1080 --streams 16 --frames 32 --steps 1 --no-extra --no-cpu-baseline --no-single > /dev/null 2>&1
for i in 1 2; do for t in 14 16 18 20 24 32; do
  for cfg in "hevc1080:--codec hevc --width 1920 --height 1080 --streams 16 --frames 32 --steps 3" "c2:--tools high_b --width 3840 --height 2160 --streams 16 --frames 24 --steps 3"; do
    n=${cfg%%:*}; a=${cfg#*:}
    JM_AMD_DEC_THREADS=$t timeout 300 python bench.py $a --no-extra --no-cpu-baseline --no-single > $P/${n}_t${t}_$i.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("$P/${n}_t${t}_$i.json")); h=d["host_cpu"]
print("$n threads $t:", d["value"], d["scaling_bound"], "cpu ms/frame", h["cpu_ms_per_frame"], "busy", h["cpus_busy"], "throttled ms", h["throttled_ms"])
PY
  done
done; done
