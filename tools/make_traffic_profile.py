"""Developer script: profiles/rNN_pmc_traffic.json from two rocprofv3 counter_collection CSVs (FETCH_SIZE pass, WRITE_SIZE pass)
of `bench.py --steps 1 --warmup 1 --frames 30 --streams 1 --no-cpu-baseline` (one picture per launch)."""
import csv, json, sys, collections, re


def per_kernel(path, counter):
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = re.sub(r"<.*", "", r['Kernel_Name'].split('(')[0].split('::')[-1])
        acc[k] += float(r['Counter_Value']) * 1024.0        # counter unit: KB
        n[k] += 1
    return {k: (acc[k] / n[k], n[k]) for k in acc}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
# optional 4th / 5th / 6th argument: the same two passes with chain launches on (k_chain holds several pictures per launch) and the bench line
# of one of those runs, from which the number of pictures that went through k_chain is taken
chain = None
if len(sys.argv) > 6:
    cf, cw = per_kernel(sys.argv[4], "FETCH_SIZE"), per_kernel(sys.argv[5], "WRITE_SIZE")
    line = json.loads(open(sys.argv[6]).read().strip().splitlines()[-1])
    kc = line["kernels"]["k_chain"]
    pics_timed = kc["launches"] * kc["pictures_per_launch"]          # timed region only; the profile also holds the warm-up pass: same stream, same count
    if "k_chain" in cf and pics_timed > 0:
        n_launch = cf["k_chain"][1]
        # the bench runs warmup + timed passes of the same stream: pictures through k_chain in the whole profile = pics_timed x (passes in profile / timed passes)
        scale = float(sys.argv[7]) if len(sys.argv) > 7 else 2.0
        pics = pics_timed * scale
        chain = {"fetch_raw": round(cf["k_chain"][0] * n_launch / pics), "write": round(cw["k_chain"][0] * cw["k_chain"][1] / pics), "launches": n_launch, "pictures": round(pics, 1)}
        chain["fetch_corrected_upper"] = 2 * chain["fetch_raw"]; chain["traffic_upper"] = chain["fetch_corrected_upper"] + chain["write"]
out = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and a separate pass --pmc WRITE_SIZE) -- python3 bench.py --steps 1 --warmup 1 --frames 30 "
              "--streams 1 --no-cpu-baseline; one picture per launch, so bytes are per 1080p picture. Counter unit KB (x1024). FETCH_SIZE is the raw value; "
              "MI355X_MICROARCH.md says it reads 1/2 of the bytes for wide (16 B/lane) coalesced streams and is uncalibrated for other widths, so "
              "fetch_corrected = 2 x raw is an upper estimate here.",
       "surface_bytes_S": 3133440, "kernels": {}}
for k in sorted(fetch):
    if not k.startswith("k_"):
        continue
    f, w = fetch[k][0], write.get(k, (0, 0))[0]
    out["kernels"][k] = {"fetch_raw": round(f), "fetch_corrected_upper": round(2 * f), "write": round(w), "traffic_upper": round(2 * f + w), "launches": fetch[k][1]}
if chain:
    out["kernels"]["k_chain"] = chain
    out["how_chain"] = "k_chain: the same two passes with chain launches on (default), bytes of all k_chain dispatches / pictures decoded through k_chain"
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
