// jmcodec_amd/csrc/hevc_kernels.h -- host-callable launcher of the HEVC kernels (hevc_kernels.hip); one call = one batch of pictures.
#pragma once
#include <hip/hip_runtime_api.h>
#include "hevc_jobs.h"

namespace jmamd {
struct HevcBatchDims { int max_pus = 0, max_tbs = 0, max_itbs = 0, max_ctb_w = 0, max_ctb_h = 0, max_w = 0, max_h = 0;
    bool any_intra = false, any_deblock = false, any_sao = false; };
// marks (optional, 4 events): before MC, after residual, after intra, after the loop filters
constexpr int kHevcIntraSegs = 8;             // workgroups per CTB row in k_hevc_intra (each walks a run of consecutive CTBs)
constexpr int kHevcProgressStride = 544 * kHevcIntraSegs;   // progress counters per picture: one per CTB row (8192 / 16 + slack) and segment
// progress: device array of n * kHevcProgressStride ints (row progress counters of k_hevc_intra, cleared by this call)
void launch_hevc_picture_batch(const HevcPicParams *d_pics, int n, const HevcBatchDims &m, int *progress, hipStream_t st, hipEvent_t *marks);
void hevc_kernels_init();
}  // namespace jmamd
