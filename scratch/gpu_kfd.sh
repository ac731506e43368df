cd $GRAFT_REPO_ROOT
ls /sys/class/kfd/kfd/proc/ 2>&1 | head; for p in /sys/class/kfd/kfd/proc/*; do echo "== $p"; ls $p 2>&1 | head -20; done 2>&1 | head -40
python - <<'PY'
import torch, os, glob, time
x = torch.zeros(1<<20, device="cuda"); torch.cuda.synchronize()
print("pid", os.getpid())
for p in glob.glob("/sys/class/kfd/kfd/proc/*"):
    print(p, os.listdir(p))
    for f in glob.glob(p + "/*"):
        if os.path.isfile(f):
            try: print("  ", os.path.basename(f), open(f).read().strip()[:80])
            except Exception as e: print("  ", os.path.basename(f), "ERR", e)
    for q in glob.glob(p + "/queues/*"):
        print("  queue", q, os.listdir(q))
        for f in glob.glob(q + "/*"):
            try: print("     ", os.path.basename(f), open(f).read().strip()[:60])
            except Exception as e: print("     ERR", e)
print(open("/sys/class/kfd/kfd/topology/nodes/1/gpu_id").read() if os.path.exists("/sys/class/kfd/kfd/topology/nodes/1/gpu_id") else "no node1")
for n in glob.glob("/sys/class/kfd/kfd/topology/nodes/*"):
    try: print(n, open(n + "/gpu_id").read().strip())
    except Exception as e: print(n, "ERR", e)
PY
