# round 4: what bounds ONE stream -- the chain launches or the caller's loop?  1 / 2 streams with frames copied out and with frames left on the device
cd $GRAFT_REPO_ROOT; P=gpurun_out/s1; mkdir -p $P
for i in 1 2 3; do for s in 1 2; do for mode in host dev; do
  X=""; [ $mode = dev ] && X="--device-output"
  timeout 300 python bench.py --no-extra --no-cpu-baseline --no-single --steps 10 --streams $s $X > $P/${mode}_s${s}_$i.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$P/${mode}_s${s}_$i.json")); k=d["kernels"]; e=d["engine"]
print("streams $s $mode:", d["value"], d["scaling_bound"], "pictures per batch", e["pictures_per_batch"], "caller wait us", e["direct_output"]["caller_wait_us_per_frame"], "pending at form", e["formation"]["pictures_waiting_per_batch_formed"], {n:(v["avg_us"],v["pictures_per_launch"]) for n,v in k.items() if n in ("k_chain","k_intra","k_deblock")})
PY
done; done; done
