"""Developer script: profiles/rNN_pmc_traffic_<codec>_<tools>_<W>x<H>.json from two rocprofv3 counter_collection CSVs (a FETCH_SIZE pass and a
WRITE_SIZE pass) of a ONE-stream bench run (one picture per launch, so bytes per launch = bytes per picture).  bench.py reads these files for
`roofline.traffic`, one per codec / tool set / picture size.

    python3 tools/make_traffic_profile.py --fetch F.csv --write W.csv --out profiles/r03_pmc_traffic_hevc_3840x2160.json \
        --width 3840 --height 2160 --command "<the bench command that was profiled>"
    (optional, H.264 chain launches: --chain-fetch / --chain-write / --chain-line <bench line of one of those runs> [--chain-scale 2.0])
"""
import argparse
import collections
import csv
import json
import re


def per_kernel(path, counter):
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = re.sub(r"<.*", "", r['Kernel_Name'].split('(')[0].split('::')[-1])
        acc[k] += float(r['Counter_Value']) * 1024.0        # counter unit: KB
        n[k] += 1
    return {k: (acc[k] / n[k], n[k]) for k in acc}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--command", default="python3 bench.py --steps 1 --warmup 1 --frames 30 --streams 1 --no-cpu-baseline")
    # environment of the profiled run, written IN FRONT of rocprofv3 in the recipe (behind the `--` it would be the program rocprofv3 tries to
    # start, and `-- env VAR=1 python3 ...` is an exec hop behind an initialised GPU, which this pool forbids under --pmc)
    ap.add_argument("--env", default="", help='e.g. "JM_AMD_DEC_CHAIN_DEPTH=1"')
    ap.add_argument("--chain-fetch")
    ap.add_argument("--chain-write")
    ap.add_argument("--chain-line")
    ap.add_argument("--chain-scale", type=float, default=2.0)
    a = ap.parse_args()

    fetch, write = per_kernel(a.fetch, "FETCH_SIZE"), per_kernel(a.write, "WRITE_SIZE")
    chain = None
    if a.chain_fetch and a.chain_write and a.chain_line:
        # the same two passes with chain launches on (k_chain holds several pictures per launch) and the bench line of one of those runs, from which
        # the number of pictures that went through k_chain is taken
        cf, cw = per_kernel(a.chain_fetch, "FETCH_SIZE"), per_kernel(a.chain_write, "WRITE_SIZE")
        line = json.loads(open(a.chain_line).read().strip().splitlines()[-1])
        kc = line["kernels"]["k_chain"]
        pics_timed = kc["launches"] * kc["pictures_per_launch"]      # timed region only; the profile also holds the warm-up pass: same stream, same count
        if "k_chain" in cf and pics_timed > 0:
            # (both chain kernels: launches that hold a picture with intra macroblocks run k_chain_i, and the line counts the pictures of both)
            tot = lambda d: sum(d[k][0] * d[k][1] for k in ("k_chain", "k_chain_i") if k in d)
            n_launch = sum(cf[k][1] for k in ("k_chain", "k_chain_i") if k in cf)
            pics = pics_timed * a.chain_scale                      # pictures through k_chain in the whole profile = timed x (passes in profile / timed passes)
            chain = {"fetch_raw": round(tot(cf) / pics), "write": round(tot(cw) / pics), "launches": n_launch,
                "pictures": round(pics, 1)}
            chain["fetch_corrected_upper"] = 2 * chain["fetch_raw"]
            chain["traffic_upper"] = chain["fetch_corrected_upper"] + chain["write"]
    mb_w, mb_h = (a.width + 15) // 16, (a.height + 15) // 16
    env = (a.env.strip() + " ") if a.env.strip() else ""
    out = {"how": f"{env}rocprofv3 --kernel-trace --pmc FETCH_SIZE (and a separate pass --pmc WRITE_SIZE) -- {a.command}; ONE stream, one picture per "
                  f"launch, so "
        f"bytes "
                  f"are per {a.width}x{a.height} picture (averaged over the launches of the kernel, i.e. over the run's picture types). Counter unit KB "
                  f"(x1024). "
                  "FETCH_SIZE is the raw value; MI355X_MICROARCH.md says it reads 1/2 of the bytes for wide (16 B/lane) coalesced streams and is uncalibrated "
                  "for "
                  "other widths, so fetch_corrected = 2 x raw is an upper estimate here.",
           "width": a.width, "height": a.height, "surface_bytes_S": mb_w * mb_h * 384, "kernels": {}}
    for k in sorted(fetch):
        if not k.startswith("k_"):
            continue
        f, w = fetch[k][0], write.get(k, (0, 0))[0]
        out["kernels"][k] = {"fetch_raw": round(f), "fetch_corrected_upper": round(2 * f), "write": round(w), "traffic_upper": round(2 * f + w),
            "launches": fetch[k][1]}
    if chain:
        out["kernels"]["k_chain"] = chain
        out["how_chain"] = "k_chain: the same two passes with chain launches on (default), bytes of all k_chain dispatches / pictures decoded through k_chain"
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
