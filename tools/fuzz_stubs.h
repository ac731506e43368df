// tools/fuzz_stubs.h -- stand-ins for the kernel launchers (never reached in parse-only mode), so that the host sources link without the device objects;
// included by the sanitizer harnesses (fuzz_host.cpp, fuzz_pushpull.cpp)
#pragma once
#include "../jmcodec_amd/csrc/kernels.h"
namespace jmamd {
bool deblock_lds_supported(int, int) { return true; }
bool intra_lds_supported(int, int) { return true; }
void launch_packout(const PackJob *, int, int, int, ihipStream_t *) { abort(); }
void launch_recon_inter(const PicParams *, int, int, bool, bool, int *, ihipStream_t *) { abort(); }
void launch_intra_lds(const PicParams *, int, int, int *, int *, ihipStream_t *) { abort(); }
void launch_recon_intra(const PicParams *, int, ihipStream_t *) { abort(); }
void launch_deblock_prep(const PicParams *, int, int, ihipStream_t *) { abort(); }
void launch_deblock_lds(const PicParams *, int, int, int *, int *, bool, ihipStream_t *) { abort(); }
bool chain_supported(int, int) { return true; }
int chain_ctl_ints() { return 1; }
int chain_tail_ints() { return 32; }
int chain_tail_head_ints() { return 32; }
int chain_tail_wait_limit() { return 15; }
void launch_chain(const PicParams *, const uint32_t *, int, bool, int *, int *, bool, ihipStream_t *) { abort(); }
int chain_band_rows() { return 16; }
int deblock_row_lag() { return 1; }
int chain_resident_workgroups(bool) { return 0; }
void launch_deblock(const PicParams *, int, ihipStream_t *) { abort(); }
}
namespace jmamd { void launch_frame_to_argb(const uint8_t *, int, int, int, uint8_t *, int, ihipStream_t *) { abort(); }
void launch_frame_to_nv12_pitch(const uint8_t *, int, int, int, uint8_t *, int, ihipStream_t *) { abort(); } }
#include "../jmcodec_amd/csrc/hevc_kernels.h"
namespace jmamd {
void launch_hevc_picture_batch(const HevcPicParams *, int, const HevcBatchDims &, int *, ihipStream_t *, ihipEvent_t **) { abort(); }
void hevc_kernels_init() {}
}
