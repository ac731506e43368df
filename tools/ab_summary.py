#!/usr/bin/env python3
"""Side-by-side summary of A/B bench runs (scratch/gpu_ab_r04.sh): <dir>/<variant>_<mode>_<i>.json -> per variant and mode the frames/s of every run and,
per kernel class, the event-timed average launch time, pictures per launch and time per picture (median over the runs)."""
import glob
import json
import os
import statistics as st
import sys


def main(d):
    out = {}
    for f in sorted(glob.glob(os.path.join(d, "*_*_*.json"))):
        name = os.path.basename(f)[:-5]
        if name in ("summary", "bench_default"):
            continue
        try:
            l = json.loads(open(f).read().strip().splitlines()[-1])
        except (IndexError, ValueError):
            continue
        variant, mode, _ = name.rsplit("_", 2)
        e = out.setdefault(f"{variant}_{mode}", {"value": [], "cpu_ms_per_frame": [], "pcie_frac": [], "kernels": {}})
        e["value"].append(l["value"])
        e["cpu_ms_per_frame"].append(l["host_cpu"]["cpu_ms_per_frame"])
        if l.get("pcie_out"):
            e["pcie_frac"].append(l["pcie_out"]["frac"])
        for k, v in l["kernels"].items():
            if v["launches"]:
                e["kernels"].setdefault(k, []).append((v["avg_us"], v["pictures_per_launch"]))
    for e in out.values():
        e["value_median"] = st.median(e["value"])
        e["kernels"] = {k: {"avg_us": round(st.median(a for a, _ in v), 1), "pictures_per_launch": round(st.median(p for _, p in v), 2),
                            "us_per_picture": round(st.median(a / max(p, 1e-9) for a, p in v), 2)} for k, v in e["kernels"].items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ab")
