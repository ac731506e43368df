// scratch/exp/glds_probe.hip -- probe (developer experiment, not product): LDS-DMA loads issued from inline asm (so that hipcc does not count them),
// counted s_waitcnt, per-wave staging slots; checks the lane -> LDS address mapping of global_load_lds_dwordx4 / _dword and times a 3-deep pipeline
// with a per-step raw barrier and conditional stores -- the structure the deblocking band kernel needs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(1))) uint8_t gbyte;
__device__ __forceinline__ void glds16(const gbyte *g, uint32_t lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void glds4_sc1(const gbyte *g, uint32_t lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_byte_addr) : "memory");
}
__global__ __launch_bounds__(256) void k(const uint8_t *src, uint8_t *dst, int n, int *bad) {
    __shared__ __align__(16) uint8_t sm[4][3][1536];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const gbyte *g = (const gbyte *)src + (size_t)blockIdx.x * 4096 + threadIdx.x * 16;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)&sm[0][0][0] + wave * 3 * 1536;
    for (int d = 0; d < 3; d++) { glds16(g + (size_t)d * 65536, base + d * 1536); glds4_sc1(g + (size_t)d * 65536 + 4, base + d * 1536 + 1024); }
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int s = 0; s < n; s++) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            uint4 v = *(const uint4 *)&sm[wave][j][lane * 16];
            uint32_t r = *(const uint32_t *)&sm[wave][j][1024 + lane * 4];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const size_t step = (size_t)s * 3 + j;
            const uint32_t *want = (const uint32_t *)(src + step * 65536 + (size_t)blockIdx.x * 4096 + threadIdx.x * 16);
            if (v.x != want[0] || v.y != want[1] || v.z != want[2] || v.w != want[3] || r != want[1]) atomicAdd(bad, 1);
            glds16(g + (step + 3) * 65536, base + j * 1536); glds4_sc1(g + (step + 3) * 65536 + 4, base + j * 1536 + 1024);
            acc.x += v.x ^ r; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            if (lane < 16 + (s & 7)) *(uint4 *)(dst + (step & 63) * 4096 + (size_t)blockIdx.x * 262144 + threadIdx.x * 16) = acc;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *(uint4 *)(dst + threadIdx.x * 16 + (size_t)blockIdx.x * 262144) = acc;
}
int main() {
    const int n = 200, blocks = 16; const size_t bytes = (size_t)(3 * n + 6) * 65536;
    std::vector<uint8_t> h(bytes); for (size_t i = 0; i < bytes; i++) h[i] = (uint8_t)(i * 2654435761u >> 13);
    uint8_t *src, *dst; int *bad; hipMalloc(&src, bytes); hipMalloc(&dst, (size_t)blocks * 262144); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) { hipEventRecord(a); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, src, dst, n, bad); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); printf("run %d: %.1f us for %d steps = %.3f us per step\n", rep, ms * 1e3, 3 * n, ms * 1e3 / (3 * n)); }
    int hb = -1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("mismatches: %d (of %d checks)  %s\n", hb, 3 * 3 * n * blocks * 256, hipGetErrorString(hipGetLastError()));
    return hb != 0;
}
