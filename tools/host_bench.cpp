// tools/host_bench.cpp -- developer tool: times the product's HOST half (NAL splitting, entropy decoding, job-list building) on one
// Annex-B stream in parse-only mode (no GPU work).  The host half is what bounds the end-to-end rate when the CPU quota is small.
//   make -C tools host_bench && tools/_build/host_bench stream.h264 [passes] [codec_type: 0 H.264, 1 HEVC]
#include "../include/jm_amd_dec.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s stream [passes] [codec_type]\n", argv[0]); return 2; }
    std::vector<unsigned char> b; { FILE *f = fopen(argv[1], "rb"); if (!f) return 2; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); b.resize(n); if (fread(b.data(), 1, n, f) != (size_t)n) return 2; fclose(f); }
    const int passes = argc > 2 ? atoi(argv[2]) : 3, codec = argc > 3 ? atoi(argv[3]) : 0;
    void *h = jm_amddec_create_handle();
    jm_amddec_set_option(h, "parse_only", 1);
    if (jm_amddec_init(codec, 1, nullptr, 0, h) != 0) { fprintf(stderr, "init failed\n"); return 1; }
    std::vector<unsigned char> out(64 << 20);
    auto t0 = std::chrono::steady_clock::now();
    long frames = jm_amddec_feed_annexb(b.data(), (long)b.size(), passes, out.data(), (int)out.size(), h);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%ld frames in %.3f s: %.1f frames/s wall, %.3f ms per frame, %.1f MB/s of bitstream\n", frames, s, frames / s, 1e3 * s / (frames > 0 ? frames : 1), passes * b.size() / s / 1e6);
    jm_amddec_deinit(h);
    return 0;
}

// kernel launchers are never reached in parse-only mode; stand-ins so that the host sources link without the device objects
#include "../jmcodec_amd/csrc/kernels.h"
#include "../jmcodec_amd/csrc/hevc_kernels.h"
namespace jmamd {
bool deblock_lds_supported(int, int) { return true; }
bool intra_lds_supported(int, int) { return true; }
void launch_packout(const PackJob *, int, int, int, ihipStream_t *) { abort(); }
void launch_recon_inter(const PicParams *, int, int, ihipStream_t *) { abort(); }
void launch_intra_lds(const PicParams *, int, int, int *, int *, ihipStream_t *) { abort(); }
void launch_recon_intra(const PicParams *, int, ihipStream_t *) { abort(); }
void launch_deblock_prep(const PicParams *, int, int, ihipStream_t *) { abort(); }
void launch_deblock_lds(const PicParams *, int, int, int *, int *, bool, ihipStream_t *) { abort(); }
bool chain_supported(int, int) { return true; }
int chain_ctl_ints() { return 1; }
void launch_chain(const PicParams *, const uint32_t *, int, int *, int *, bool, ihipStream_t *) { abort(); }
int chain_band_rows() { return 16; }
void launch_deblock(const PicParams *, int, ihipStream_t *) { abort(); }
void launch_frame_to_argb(const uint8_t *, int, int, int, uint8_t *, int, ihipStream_t *) { abort(); }
void launch_frame_to_nv12_pitch(const uint8_t *, int, int, int, uint8_t *, int, ihipStream_t *) { abort(); }
void launch_hevc_picture_batch(const HevcPicParams *, int, const HevcBatchDims &, int *, ihipStream_t *, ihipEvent_t **) { abort(); }
void hevc_kernels_init() {}
}
