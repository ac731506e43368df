#include <hip/hip_runtime.h>
#include "jobs.h"
#include "kernels.h"
#include "kernel_common.h"
#include "chain_common.h"
#include "recon_device.h"
namespace jmamd {
template <bool CHAIN, bool COH, bool BI>
__global__ __launch_bounds__(256) void k_probe(const PicParams *pics, int *ctl, int *err, const uint32_t *groups) {
    const int g = (int)blockIdx.x >> 1, rem = (int)blockIdx.x & 1;
    const uint32_t entry = groups[g];
    const PicParams &pp = pics[entry >> 16];
    const ChainView cv{ctl, err};
    const int row = (int)(entry & 0x7fffu) >> 5, seg = (int)entry & 31;
    const int x = seg * 8 + rem * 4 + (int)(threadIdx.x >> 6);
    const bool valid = row < pp.mb_h && x < pp.mb_w;
    if (row >= pp.mb_h || seg * 8 + rem * 4 >= pp.mb_w) return;
    __shared__ ReconLds sm;
    recon_inter_wave<CHAIN, COH, BI, false>(pp, valid ? row * pp.mb_w + x : 0, valid, sm, cv);
}
template __global__ void k_probe<true, true, false>(const PicParams *, int *, int *, const uint32_t *);
template __global__ void k_probe<true, true, true>(const PicParams *, int *, int *, const uint32_t *);
}
