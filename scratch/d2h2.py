import torch, time
n = 3110400
s = torch.cuda.Stream()
for nbuf in (4, 64, 256, 900):
    src = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(min(nbuf, 64))]
    dst = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
    torch.cuda.synchronize()
    for rep in range(2):
        t = time.perf_counter(); k = 0
        with torch.cuda.stream(s):
            for it in range(1200):
                dst[k % nbuf].copy_(src[k % len(src)], non_blocking=True); k += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(nbuf, "pinned buffers:", k * n / dt / 1e9, "GB/s")
    del dst
