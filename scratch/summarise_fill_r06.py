#!/usr/bin/env python3
"""Round 6: means per configuration of the fill-linger experiments (gpurun_out/<dir>/<leg>_<cfg>_<run>.json) -> text for profiles/r06_copy_streams.txt"""
import json, glob, os, sys, statistics as st, collections
def rows(d):
    g = collections.defaultdict(list)
    for f in sorted(glob.glob(f'gpurun_out/{d}/*.json')):
        try: x = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception: continue
        leg, cfg = os.path.basename(f)[:-5].rsplit('_', 1)[0].split('_', 1)
        g[(leg, cfg)].append(x)
    out = []
    for (leg, cfg), xs in sorted(g.items()):
        o = [x["engine"]["lanes"].get("ordinary", {}) for x in xs]; it = [x["engine"]["lanes"].get("intra", {}) for x in xs]
        m = lambda v: st.mean(v) if v else float('nan')
        out.append(f"{leg:6s} {cfg:8s} runs {len(xs)}  frames/s {m([x['value'] for x in xs]):8.0f} ({min(x['value'] for x in xs):.0f}-{max(x['value'] for x in xs):.0f})  "
                   f"ordinary lane {m([a.get('pictures_per_batch', 0) for a in o]):5.2f} pictures per batch, busy {m([a.get('busy_frac', 0) for a in o]):.2f}  "
                   f"intra lane {m([a.get('batches', 0) for a in it]):4.0f} batches x {m([a.get('pictures_per_batch', 0) for a in it]):.2f}  "
                   f"k_deblock {m([x['roofline']['frac'] for x in xs if x['roofline']['kernel'] == 'k_deblock']):.4f} of the roof  "
                   f"pcie {m([(x.get('pcie_out') or {}).get('frac') or 0 for x in xs]):.3f}  all bit-exact {all(x['bit_exact'] for x in xs)}")
    return out
if __name__ == "__main__":
    for d in sys.argv[1:]:
        print(f"## gpurun_out/{d}"); print("\n".join(rows(d)))
