// jmcodec_amd/csrc/jm_amd_dec.cpp -- C ABI (include/jm_amd_dec.h) over jmamd::Decoder.
#include "../../include/jm_amd_dec.h"
#include "decoder.h"
#include "kernels.h"
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <vector>

using jmamd::Decoder;

// The engine drives several HIP streams (per lane: decode + pack-out, plus one copy stream) that must map to distinct
// hardware queues to run concurrently; ROCm's default is 4 queues per process.  Must be set before the runtime initialises.
__attribute__((constructor)) static void jm_amddec_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

// The host half is compiled for BMI1 / BMI2 / LZCNT (Makefile: the arithmetic decoder's variable shifts and leading-zero counts; every x86 host an
// MI355X is sold in has them).  A CPU without them gets told so before the first such instruction runs, instead of an illegal-instruction trap.
#if defined(__BMI2__) || defined(__BMI__) || defined(__LZCNT__)
__attribute__((constructor(101), target("no-bmi,no-bmi2,no-lzcnt"))) static void jm_amddec_cpu_check() {
    __builtin_cpu_init();
    if (!__builtin_cpu_supports("bmi") || !__builtin_cpu_supports("bmi2") || !__builtin_cpu_supports("lzcnt")) {
        fputs("jm_amd_dec: this build needs a CPU with BMI1, BMI2 and LZCNT (rebuild jmcodec_amd/csrc with HOST_ISA= for older hosts)\n", stderr);
        abort();
    }
}
#endif

#define D(h) (reinterpret_cast<Decoder *>(h))

// No C++ exception may cross the C ABI (a hostile stream must not be able to abort the host application through an allocation
// failure or a container bound): the entry points that run parser code convert them into the API's error return.
template <class F> static int guarded(jm_amddec_handle h, F &&f) {
    try { return f(); }
    catch (const std::exception &e) { D(h)->api_exception(e.what()); }
    catch (...) { D(h)->api_exception("unknown exception"); }
    return -1;
}

extern "C" {

__attribute__((visibility("default"))) jm_amddec_handle jm_amddec_create_handle(void) { return new Decoder(); }
__attribute__((visibility("default"))) int jm_amddec_init(int codec_type, int out_fmt, char *extra, int len, jm_amddec_handle h) {
    if (!h) return -1;
    return guarded(h, [&] { return D(h)->init(codec_type, out_fmt, reinterpret_cast<const uint8_t *>(extra), len); });
}
__attribute__((visibility("default"))) int jm_amddec_deinit(jm_amddec_handle h) { delete D(h); return 0; }
__attribute__((visibility("default"))) int jm_amddec_decode_frame(unsigned char *in_buf, int n, int *got, jm_amddec_handle h) {
    int dummy = 0;
    if (!h) return -1;
    return guarded(h, [&] { return D(h)->decode(in_buf, n, got ? got : &dummy); });
}
__attribute__((visibility("default"))) int jm_amddec_poll_frame(int *got, jm_amddec_handle h) {
    int dummy = 0;
    if (!h) return -1;
    return guarded(h, [&] { return D(h)->poll(got ? got : &dummy); });
}
__attribute__((visibility("default"))) int jm_amddec_wait_frame(int *got, int timeout_us, jm_amddec_handle h) {
    int dummy = 0;
    if (!h) return -1;
    return guarded(h, [&] { return D(h)->poll(got ? got : &dummy, timeout_us); });
}
__attribute__((visibility("default"))) int jm_amddec_push_data(unsigned char *in_buf, int n, jm_amddec_handle h) {
    if (!h) return -1;
    return guarded(h, [&] { return D(h)->push(in_buf, n); });
}
__attribute__((visibility("default"))) int jm_amddec_push_eos(jm_amddec_handle h) {
    if (!h) return -1;
    return guarded(h, [&] { return D(h)->push_eos(); });
}
__attribute__((visibility("default"))) int jm_amddec_output_frame(unsigned char *out, int *out_len, jm_amddec_handle h) {
    if (!h || !out || !out_len) return -1;
    return guarded(h, [&] { return D(h)->output(out, out_len); });
}
__attribute__((visibility("default"))) int jm_amddec_stream_info(int *w, int *hh, jm_amddec_handle h) { return D(h)->stream_info(w, hh); }
__attribute__((visibility("default"))) void jm_amddec_set_eof(int e, jm_amddec_handle h) { D(h)->set_eof(e != 0); }
__attribute__((visibility("default"))) int jm_amddec_is_exit(jm_amddec_handle h) { return D(h)->is_exit() ? 1 : 0; }
__attribute__((visibility("default"))) char *jm_amddec_show_dec_info(jm_amddec_handle h) { return D(h)->info(); }
__attribute__((visibility("default"))) int jm_amddec_is_hw_support(void) {      // nvdec_cuda_hw_support: device count > 0 (nv_dec.cpp:188-200)
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}
__attribute__((visibility("default"))) int jm_amddec_set_option(jm_amddec_handle h, const char *key, long long v) { return D(h)->set_option(key, v); }
__attribute__((visibility("default"))) long long jm_amddec_get_stat(jm_amddec_handle h, const char *key) { return D(h)->get_stat(key); }
__attribute__((visibility("default"))) const char *jm_amddec_last_error(jm_amddec_handle h) { return D(h)->last_error(); }
__attribute__((visibility("default"))) int jm_amddec_output_frame_device(void **dev, int *len, jm_amddec_handle h) {
    return (h && dev && len) ? D(h)->output_device(dev, len) : -1; }
__attribute__((visibility("default"))) int jm_amddec_output_argb_device(void *dev_dst, int pitch, jm_amddec_handle h) {
    return (h && dev_dst) ? D(h)->output_argb_device(dev_dst, pitch) : -1; }
__attribute__((visibility("default"))) int jm_amddec_output_nv12_pitch_device(void *dev_dst, int pitch, jm_amddec_handle h) {
    return (h && dev_dst) ? D(h)->output_nv12_pitch_device(dev_dst, pitch) : -1; }
__attribute__((visibility("default"))) int jm_amddec_i420_to_nv12_device(const void *d_src, int width, int height, int src_fmt, void *d_dst, int pitch,
    void *stream) {
    if (!d_src || !d_dst || width <= 0 || height <= 0 || (width & 1) || (height & 1) || pitch < width || (src_fmt != 0 && src_fmt != 1)) return -1;
    jmamd::launch_frame_to_nv12_pitch((const uint8_t *)d_src, width, height, src_fmt, (uint8_t *)d_dst, pitch, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
__attribute__((visibility("default"))) int jm_amddec_packout_device(const void *src, int pitch, int w, int hgt, int fmt, void *dst, void *stream) {
    jmamd::PackJob job{static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), pitch, pitch * hgt, w, hgt, fmt, 0};
    jmamd::PackJob *d_job = nullptr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMalloc((void **)&d_job, sizeof job) != hipSuccess) return -1;
    hipMemcpyAsync(d_job, &job, sizeof job, hipMemcpyHostToDevice, st);
    jmamd::launch_packout(d_job, 1, w, hgt, st);
    hipError_t e = hipGetLastError();
    hipStreamSynchronize(st);
    hipFree(d_job);
    return e == hipSuccess ? 0 : -(int)e;
}

__attribute__((visibility("default"))) long jm_amddec_feed_annexb(const unsigned char *buf, long len, int passes, unsigned char *out, int out_cap,
    jm_amddec_handle h) {
    if (!h || !buf || len < 4) return -1;       // out == NULL: frames stay on the device (jm_amddec_output_frame_device), nothing is copied
    // NAL boundaries as find_nalu sees them: a start code is 00 00 01, or 00 00 00 01 (then the NAL starts one byte earlier)
    std::vector<long> starts;
    for (long i = 0; i + 3 <= len; i++) if (buf[i] == 0 && buf[i + 1] == 0 && buf[i + 2] == 1) { long s0 = (i > 0 && buf[i - 1] == 0) ? i - 1 : i;
        if (starts.empty() || s0 > starts.back()) starts.push_back(s0); i += 2; }
    if (starts.empty()) return -1;
    long frames = 0;
    for (int p = 0; p < passes; p++)
        for (size_t k = 0; k < starts.size(); k++) {
            const long b = starts[k], e = k + 1 < starts.size() ? starts[k + 1] : len;
            int got = 0;
            if (guarded(h, [&] { return D(h)->decode(buf + b, (int)(e - b), &got); }) != 0) return -2;
            if (got == 1) {
                if (out) { int n = out_cap; if (D(h)->output(out, &n) > 0) frames++; }
                else { void *dev = nullptr; int n = 0; if (D(h)->output_device(&dev, &n) > 0) frames++; }
            }
        }
    return frames;
}

}  // extern "C"
