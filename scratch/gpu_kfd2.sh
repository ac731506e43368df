cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, os, glob
x = torch.zeros(1<<20, device="cuda"); torch.cuda.synchronize()
me=os.getpid(); print("me", me)
mine=set()
for q in glob.glob(f"/sys/class/kfd/kfd/proc/{me}/queues/*/gpuid"): mine.add(open(q).read().strip())
print("my gpuids", mine)
for p in glob.glob("/sys/class/kfd/kfd/proc/*"):
    pid=os.path.basename(p)
    qs=[]
    for q in glob.glob(p+"/queues/*"):
        try: qs.append((open(q+"/gpuid").read().strip(), open(q+"/type").read().strip(), open(q+"/size").read().strip()))
        except Exception as e: qs.append(("ERR",str(e)))
    try: cmd=open(f"/proc/{pid}/cmdline").read().replace("\0"," ")[:80]
    except Exception as e: cmd="(no /proc entry: %s)"%type(e).__name__
    print(pid, "queues", len(qs), [q for q in qs if q[0] in mine], cmd)
PY
