// tools/sdma_probe.cpp -- developer tool: which SDMA engines can move a packed 1080p frame (3.1 MB) from device memory to page-locked host memory,
// and how fast -- one engine at a time and two at once (hsa_amd_memory_async_copy_on_engine).  The HIP runtime sends every device-to-host copy to ONE
// engine; the product's "direct" output route needs to know whether a second one raises the rate over the PCIe link.
//   make -C tools sdma_probe && timeout 60 tools/_build/sdma_probe
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

static hsa_agent_t g_gpu, g_cpu; static bool g_have_gpu = false, g_have_cpu = false;
static hsa_status_t on_agent(hsa_agent_t a, void *) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t n = 1920 * 1080 * 3 / 2; const int reps = 200;
    void *dev = nullptr; if (hipMalloc(&dev, n * 4) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    hipMemset(dev, 7, n * 4); hipDeviceSynchronize();
    if (hsa_init() != HSA_STATUS_SUCCESS) return 1;
    hsa_iterate_agents(on_agent, nullptr);
    if (!g_have_gpu || !g_have_cpu) return 1;
    uint32_t mask = 0, pref = 0;
    hsa_status_t s1 = hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &mask), s2 = hsa_amd_memory_get_preferred_copy_engine(g_cpu, g_gpu, &pref);
    printf("device->host engines: status 0x%x (rc %d), preferred 0x%x (rc %d)\n", mask, (int)s1, pref, (int)s2);
    std::vector<void *> host(4), hostg(4);
    for (int i = 0; i < 4; i++) {
        if (posix_memalign(&host[i], 4096, n + 4096)) return 1;
        hsa_status_t st = hsa_amd_memory_lock(host[i], n, &g_gpu, 1, &hostg[i]);
        if (st != HSA_STATUS_SUCCESS) { printf("lock failed %d\n", (int)st); return 1; }
    }
    hsa_signal_t sig[4]; for (auto &s : sig) hsa_signal_create(1, 0, nullptr, &s);
    auto copy = [&](int slot, uint32_t engine) {
        hsa_signal_store_relaxed(sig[slot], 1);
        return engine ? hsa_amd_memory_async_copy_on_engine(hostg[slot], g_cpu, (char *)dev + n * slot, g_gpu, n, 0, nullptr, sig[slot],
            (hsa_amd_sdma_engine_id_t)engine, false)
                      : hsa_amd_memory_async_copy(hostg[slot], g_cpu, (char *)dev + n * slot, g_gpu, n, 0, nullptr, sig[slot]);
    };
    auto wait = [&](int slot) { while (hsa_signal_wait_scacquire(sig[slot], HSA_SIGNAL_CONDITION_LT, 1, 1000000000ull, HSA_WAIT_STATE_BLOCKED) >= 1) {} };
    // one engine at a time (0 = the runtime's own choice), two copies in flight
    for (uint32_t e = 0; e <= 0x80; e = e ? e << 1 : 1) {
        if (e && !(mask & e)) continue;
        hsa_status_t st = copy(0, e); if (st != HSA_STATUS_SUCCESS) { printf("engine 0x%x: rc %d\n", e, (int)st); continue; }
        wait(0);
        double t0 = now();
        copy(0, e); 
        for (int i = 1; i < reps; i++) { copy(i & 1, e); wait((i - 1) & 1); }
        wait((reps - 1) & 1);
        double dt = now() - t0;
        printf("engine 0x%02x alone: %.1f us per frame, %.1f GB/s  (first byte %d)\n", e, 1e6 * dt / reps, n * reps / dt / 1e9, ((unsigned char *)host[0])[0]);
    }
    // two engines at once
    std::vector<uint32_t> es; for (uint32_t e = 1; e <= 0x80; e <<= 1) if (mask & e) es.push_back(e);
    for (size_t a = 0; a < es.size(); a++) for (size_t b = a + 1; b < es.size(); b++) {
        double t0 = now();
        copy(0, es[a]); copy(1, es[b]);
        for (int i = 1; i < reps / 2; i++) { copy(2, es[a]); copy(3, es[b]); wait(0); wait(1); copy(0, es[a]); copy(1, es[b]); wait(2); wait(3); }
        wait(0); wait(1);
        double dt = now() - t0; const int frames = 2 + 4 * (reps / 2 - 1);
        printf("engines 0x%02x + 0x%02x: %.1f us per frame, %.1f GB/s\n", es[a], es[b], 1e6 * dt / frames, n * (double)frames / dt / 1e9);
    }
    // three engines at once (what the product's direct output route does): 0x2, 0x4, 0x8 in turn, four copies in flight
    if ((mask & 0xe) == 0xe) {
        const uint32_t e3[3] = {0x2, 0x4, 0x8};
        double t0 = now();
        for (int s2 = 0; s2 < 4; s2++) copy(s2, e3[s2 % 3]);
        int issued = 4;
        for (int i = 0; i < 3 * reps; i++) { const int s2 = i & 3; wait(s2); if (issued < 3 * reps + 4) { copy(s2, e3[issued % 3]); issued++; } }
        for (int s2 = 0; s2 < 4; s2++) wait(s2);
        double dt = now() - t0;
        printf("engines 0x02 + 0x04 + 0x08 in turn, 4 in flight: %.1f us per frame, %.1f GB/s\n", 1e6 * dt / issued, n * (double)issued / dt / 1e9);
    }
    // what page-locking the caller's buffer per call would cost (instead of keeping it locked between calls)
    {
        void *p = nullptr, *ap = nullptr; if (posix_memalign(&p, 4096, n)) return 1;
        memset(p, 1, n);
        double t0 = now();
        for (int i = 0; i < reps; i++) { hsa_amd_memory_lock(p, n, &g_gpu, 1, &ap); hsa_amd_memory_unlock(p); }
        printf("hsa_amd_memory_lock + unlock of %zu bytes: %.1f us per pair\n", n, 1e6 * (now() - t0) / reps);
        t0 = now();
        for (int i = 0; i < reps; i++) { hsa_amd_memory_lock(p, n, &g_gpu, 1, &ap); copy(0, 0x2); hsa_signal_store_relaxed(sig[0], 1);
            hsa_amd_memory_async_copy_on_engine(ap, g_cpu, dev, g_gpu, n, 0, nullptr, sig[0], (hsa_amd_sdma_engine_id_t)0x2, false); wait(0);
            hsa_amd_memory_unlock(p); }
        printf("lock + copy + unlock: %.1f us per frame\n", 1e6 * (now() - t0) / reps);
    }
    // the HIP way for comparison
    void *ph = nullptr; hipHostMalloc(&ph, n * 2, hipHostMallocDefault); hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipMemcpyAsync(ph, dev, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
    double t0 = now();
    for (int i = 0; i < reps; i++) hipMemcpyAsync((char *)ph + n * (i & 1), dev, n, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    double dt = now() - t0;
    printf("hipMemcpyAsync stream: %.1f us per frame, %.1f GB/s\n", 1e6 * dt / reps, n * reps / dt / 1e9);
    return 0;
}
