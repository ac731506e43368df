#!/usr/bin/env python3
"""Round 6: copy the final call's outputs (scratch/gpu_final_r06.sh -> gpurun_out/p6, f6, sq5) into profiles/ under their tracked names."""
import json, os, shutil, glob
P = "profiles"
def cp(src, dst):
    if os.path.exists(src): shutil.copyfile(src, os.path.join(P, dst)); print("copied", src, "->", dst)
    else: print("MISSING", src)
for f in sorted(glob.glob("gpurun_out/p6/r06_*")):
    cp(f, os.path.basename(f))
for n in ("r06_bench_s1.json", "r06_bench_s8.json", "r06_bench_high.json", "r06_bench_high_b.json"):
    cp("gpurun_out/f6/" + n, n)
for c in ("c1", "c2", "c3"):
    src = f"gpurun_out/sq5/final_{c}.json"
    if not os.path.exists(src): print("MISSING", src); continue
    old = json.load(open(f"{P}/r05_sq_counters_{c}.json"))
    out = {"round5_final": old["final"], "round6_final": json.load(open(src))}
    json.dump(out, open(f"{P}/r06_sq_counters_{c}.json", "w"), indent=1); print("wrote", f"r06_sq_counters_{c}.json")
# GPU suite tail + sweeps
with open(f"{P}/r06_gputests_final.log", "w") as o:
    o.write("# tail of `pytest tests -m gpu -q -rs` in the round's final call (scratch/gpu_final_r06.sh)\n")
    o.write("".join(open("gpurun_out/f6/gputests.log").readlines()[-12:]))
with open(f"{P}/r06_gpu_sweep.log", "w") as o:
    o.write("# tools/gpu_sweep.py in the round's final call: 400 (seed 101) + 400 (seed 102) + 60 big (seed 103) random configurations per codec\n")
    for n in ("sweep_a", "sweep_b", "sweep_big"):
        L = open(f"gpurun_out/f6/{n}.log").readlines()
        o.write(f"## {n}\n" + "".join(L[-6:]))
