import torch, time
n = 3110400
for ns in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    src = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(ns * 4)]
    dst = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(ns * 4)]
    torch.cuda.synchronize()
    for rep in range(2):
        t = time.perf_counter()
        k = 0
        for it in range(200):
            for s in range(ns):
                with torch.cuda.stream(streams[s]):
                    dst[k % len(dst)].copy_(src[k % len(src)], non_blocking=True); k += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(ns, "streams:", k * n / dt / 1e9, "GB/s")
