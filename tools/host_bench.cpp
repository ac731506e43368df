// tools/host_bench.cpp -- developer tool: times the product's HOST half (NAL splitting, entropy decoding, job-list building) on one
// Annex-B stream in parse-only mode (no GPU work).  The host half is what bounds the end-to-end rate when the CPU quota is small.
//   make -C tools host_bench && tools/_build/host_bench stream.h264 [passes] [codec_type: 0 H.264, 1 HEVC]
#include "../include/jm_amd_dec.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <csignal>
#include <cstring>
#include <sys/time.h>
#include <ucontext.h>
#include <time.h>
#include <pthread.h>

// JM_HOST_BENCH_PROF=file: a sampling profile of the host half (the image has no perf / gprof-capable toolchain): SIGPROF every 50 us of wall time, the
// interrupted program counter is recorded and written as `offset-in-binary` lines for llvm-addr2line / llvm-symbolizer.
static unsigned long *g_pcs; static volatile long g_npc; static const long kMaxPc = 1 << 22;
static void on_prof(int, siginfo_t *, void *uc) { long i = __atomic_fetch_add(&g_npc, 1, __ATOMIC_RELAXED);
    if (i < kMaxPc) g_pcs[i] = (unsigned long)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP]; }
static void prof_start() {
    g_pcs = new unsigned long[kMaxPc];
    struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART; sigaction(SIGPROF, &sa, nullptr);
    // a wall-clock timer (ITIMER_PROF only ticks at CONFIG_HZ); the signal is blocked in the calling thread, so it interrupts the parse workers,
    // which do all the work in parse-only mode with JM_AMD_DEC_THREADS=1
    timer_t t; struct sigevent se; memset(&se, 0, sizeof se); se.sigev_notify = SIGEV_SIGNAL; se.sigev_signo = SIGPROF; timer_create(CLOCK_MONOTONIC, &se, &t);
    struct itimerspec its = {{0, 50000}, {0, 50000}}; timer_settime(t, 0, &its, nullptr);
    sigset_t m; sigemptyset(&m); sigaddset(&m, SIGPROF); pthread_sigmask(SIG_BLOCK, &m, nullptr);
}
static void prof_stop(const char *path) {
    signal(SIGPROF, SIG_IGN);
    unsigned long lo = 0, hi = 0;                                           // the executable's own text mapping
    if (FILE *m = fopen("/proc/self/maps", "r")) { char ln[512]; while (fgets(ln, sizeof ln, m)) { unsigned long a, b, off; char perm[8];
        if (sscanf(ln, "%lx-%lx %7s %lx", &a, &b, perm, &off) == 4 && strstr(ln, "host_bench")) { if (!lo || a - off < lo) lo = a - off; if (b > hi) hi = b; }
        } fclose(m); }
    if (FILE *f = fopen(path, "w")) { long n = g_npc < kMaxPc ? g_npc : kMaxPc; for (long i = 0; i < n; i++) if (g_pcs[i] >= lo && g_pcs[i] < hi) fprintf(f,
        "0x%lx\n", g_pcs[i] - lo); else fprintf(f, "other 0x%lx\n", g_pcs[i]); fclose(f); }
    // (the samples outside the executable -- libc, libstdc++, the kernel's entry stubs -- carry their absolute address; the mappings go beside them)
    { char mp[600]; snprintf(mp, sizeof mp, "%s.maps", path); FILE *o = fopen(mp, "w"), *m = fopen("/proc/self/maps", "r"); char ln[512];
      if (o && m) while (fgets(ln, sizeof ln, m)) if (strstr(ln, " r-xp ")) fputs(ln, o);
      if (o) fclose(o); if (m) fclose(m); }
}

#ifdef JM_COUNT_BINS
namespace jmamd { long g_cabac_bins = 0; }
#endif
int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s stream [passes] [codec_type]\n", argv[0]); return 2; }
    std::vector<unsigned char> b;
    { FILE *f = fopen(argv[1], "rb"); if (!f) return 2; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); b.resize(n);
        if (fread(b.data(), 1, n, f) != (size_t)n) return 2; fclose(f); }
    const int passes = argc > 2 ? atoi(argv[2]) : 3, codec = argc > 3 ? atoi(argv[3]) : 0;
    void *h = jm_amddec_create_handle();
    jm_amddec_set_option(h, "parse_only", 1);
    if (jm_amddec_init(codec, 1, nullptr, 0, h) != 0) { fprintf(stderr, "init failed\n"); return 1; }
    std::vector<unsigned char> out(64 << 20);
    const char *prof = getenv("JM_HOST_BENCH_PROF");
    // (the first pass creates the parse workers: they must not inherit the blocked signal)
    if (prof) { jm_amddec_feed_annexb(b.data(), (long)b.size(), 1, out.data(), (int)out.size(), h); prof_start(); }
    auto t0 = std::chrono::steady_clock::now();
    long frames = jm_amddec_feed_annexb(b.data(), (long)b.size(), passes, out.data(), (int)out.size(), h);
    // drain inside the timed region: every picture fed has then been parsed (the pipeline holds up to two dozen per handle)
    for (int i = 0, got = 0; i < 100000 && !jm_amddec_is_exit(h); i++) { jm_amddec_decode_frame(nullptr, 0, &got, h); if (got) { int n = (int)out.size();
        if (jm_amddec_output_frame(out.data(), &n, h) > 0) frames++; } }
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (prof) prof_stop(prof);
    // per PICTURE PARSED (the stat), not per frame returned: a reordering DPB holds frames back across the passes
    long pics = (long)jm_amddec_get_stat(h, "pictures"); if (pics <= 0) pics = frames > 0 ? frames : 1;
    if (prof) pics -= pics / (passes + 1);            // (the profiler's warm-up pass is not in the timed region)
#ifdef JM_COUNT_BINS
    printf("%.0f bins per picture (all passes; JM_AMD_DEC_THREADS=1 for a meaningful count)\n", (double)jmamd::g_cabac_bins / pics);
#endif
    printf("%ld pictures (%ld frames returned) in %.3f s: %.1f pictures/s wall, %.3f ms per picture, %.1f MB/s of bitstream\n", pics, frames, s, pics / s,
        1e3 * s / pics, passes * b.size() / s / 1e6);
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
void launch_frame_to_argb(const uint8_t *, int, int, int, uint8_t *, int, ihipStream_t *) { abort(); }
void launch_frame_to_nv12_pitch(const uint8_t *, int, int, int, uint8_t *, int, ihipStream_t *) { abort(); }
void launch_hevc_picture_batch(const HevcPicParams *, int, const HevcBatchDims &, int *, ihipStream_t *, ihipEvent_t **) { abort(); }
void hevc_kernels_init() {}
}
