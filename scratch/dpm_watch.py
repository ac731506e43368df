"""Developer script (round 6): log every change of the GPU's clock / power levels with CLOCK_MONOTONIC time stamps while a bench run goes on, to set beside the
clock gaps the chain launches' waits report (JM_AMD_DEC_VERBOSE prints CLOCK_MONOTONIC too).  usage: python scratch/dpm_watch.py out.txt seconds"""
import glob, sys, time
out, secs = sys.argv[1], float(sys.argv[2])
dev = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
base = dev[0].rsplit("/", 1)[0] if dev else None
names = ["pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "power_dpm_force_performance_level", "gpu_busy_percent"]
def cur(n):
    try:
        t = open(f"{base}/{n}").read()
    except OSError:
        return None
    if n.startswith("pp_dpm"):
        star = [l for l in t.splitlines() if l.rstrip().endswith("*")]
        return star[0].strip() if star else t.strip().replace("\n", " | ")
    return t.strip()
with open(out, "w") as f:
    f.write(f"# base {base}\n")
    if base is None:
        sys.exit(0)
    last = {}
    t_end = time.monotonic() + secs
    n = 0
    while time.monotonic() < t_end:
        for nm in names[:4]:
            v = cur(nm)
            if v != last.get(nm):
                f.write(f"{time.monotonic():.3f} {nm}: {last.get(nm)} -> {v}\n"); f.flush()
                last[nm] = v
        n += 1
        time.sleep(0.002)
    f.write(f"# {n} polls\n")
