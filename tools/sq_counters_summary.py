"""Developer script: profiles/rNN_sq_counters_<config>.json from rocprofv3 --pmc counter_collection CSVs of ONE-stream bench runs with chain launches off
(one picture per launch, so a kernel's counters per launch are its counters per picture).  One CSV per counter group (separate passes, as the
microarchitecture guide prescribes); values are summed over the XCDs / SEs the way rocprofv3 reports them and averaged over the launches of a kernel.

    python3 tools/sq_counters_summary.py --out profiles/r05_sq_counters_c1.json --config "C1 ..." --command "<bench command>" [--tag before] a.csv b.csv ...
"""
import argparse
import collections
import csv
import json
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--config", required=True)
    ap.add_argument("--command", required=True)
    ap.add_argument("--tag", default="")
    ap.add_argument("--merge", action="store_true", help="add this set under its tag to an existing file instead of replacing it")
    ap.add_argument("csvs", nargs="+")
    a = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    dur = collections.defaultdict(float)
    ndur = collections.Counter()
    seen = set()
    for path in a.csvs:
        for r in csv.DictReader(open(path)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].split("::")[-1]
            if not k.startswith("k_"):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
            key = (path, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                ndur[k] += 1
    kernels = {}
    for k in sorted(acc):
        d = {c: round(v / cnt[(k, c)]) for c, v in sorted(acc[k].items())}
        d["launches"] = max(cnt[(k, c)] for c in acc[k])
        d["avg_us_under_pmc"] = round(dur[k] / max(1, ndur[k]) / 1000.0, 2)
        if "SQ_WAIT_ANY" in d and d.get("SQ_WAVE_CYCLES"):
            d["wait_share"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3)
        kernels[k] = d
    how = ("rocprofv3 --kernel-trace --pmc <group> per pass (no other trace domain); one stream, JM_AMD_DEC_CHAIN_DEPTH=1: one picture per launch, "
           "so every figure is per picture and launch, averaged over the kernel's launches (all picture types of the run).  SQ_INSTS_* count "
           "wave-instructions; SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_BUSY_CYCLES are summed over the shader engines as rocprofv3 reports them.")
    entry = {"config": a.config, "command": a.command, "how": how, "kernels": kernels}
    out = {}
    if a.merge:
        try:
            out = json.load(open(a.out))
        except OSError:
            out = {}
    if a.tag:
        out[a.tag] = entry
    else:
        out = entry
    json.dump(out, open(a.out, "w"), indent=1)
    for k, d in kernels.items():
        print(k, d)


if __name__ == "__main__":
    main()
