// tools/test_amd_dec.cpp -- the native command-line harness: the counterpart of the reference's
// test_nv_dec (/root/reference/test_nv_dec/test_nv_dec.cpp:98-268) for libjm_amd_dec.so.
//
//   test_amd_dec <in.h264|in.h265> [out.yuv] [--codec 0|1] [--fmt 0|1] [--loops N] [--chunk BYTES]
//
// Same protocol as the reference harness: read the file through a sliding buffer, cut it into NAL units at start codes, one
// jm_nvdec_decode_frame call per NAL unit, jm_nvdec_output_frame whenever got_frame == 1, then (NULL, 0) calls until
// jm_nvdec_is_exit; finally print jm_nvdec_show_dec_info and the NAL count.  It is compiled against the ten declarations of the
// reference header (nv_dec/jm_nv_dec.h:20-88, repeated below because /root/reference does not exist on the GPU box) and links the
// Itanium-mangled jm_nvdec_* symbols the library exports -- i.e. exactly what a rebuilt test_nv_dec would bind.
// Differences from the reference harness: file names come from argv (the reference hard-codes f:\ paths), the YUV really is
// written when an output path is given (the reference has its fwrite commented out, :221,243), --loops re-feeds the input
// (throughput measurement on short clips) and there is no Win32 kbhit() exit.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef void *handle_nvdec;                                                                    // jm_nv_dec.h:20
handle_nvdec jm_nvdec_create_handle();                                                         // :27
int jm_nvdec_init(int codec_type, int out_fmt, char *extra_data, int len, handle_nvdec handle); // :39
int jm_nvdec_deinit(handle_nvdec handle);                                                      // :47
int jm_nvdec_decode_frame(unsigned char *in_buf, int in_data_len, int *got_frame, handle_nvdec handle);   // :58
int jm_nvdec_output_frame(unsigned char *out_buf, int *out_len, handle_nvdec handle);          // :68
int jm_nvdec_stream_info(int *disp_width, int *disp_height, handle_nvdec handle);              // :79
void jm_nvdec_set_eof(bool is_eof, handle_nvdec handle);                                       // :82
bool jm_nvdec_is_exit(handle_nvdec handle);                                                    // :84
char *jm_nvdec_show_dec_info(handle_nvdec handle);                                             // :86
bool jm_nvdec_is_hw_support();                                                                 // :88

// end of the NAL unit that starts at buf (which begins with a start code): offset of the next start code, or -1 when the
// buffer holds no further start code (the NAL may continue in data not read yet)
static long next_start_code(const unsigned char *buf, long len) {
    for (long i = 3; i + 3 <= len; i++)
        if (buf[i] == 0 && buf[i + 1] == 0 && buf[i + 2] == 1) return (i > 3 && buf[i - 1] == 0) ? i - 1 : i;
    return -1;
}

int main(int argc, char **argv) {
    const char *in_path = nullptr, *out_path = nullptr;
    int codec = 0, fmt = 1, loops = 1; long chunk = 4 << 20; bool hw_check = true;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--codec") && i + 1 < argc) codec = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--fmt") && i + 1 < argc) fmt = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--loops") && i + 1 < argc) loops = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--chunk") && i + 1 < argc) chunk = atol(argv[++i]);
        // host-side tests: with JM_AMD_DEC_PARSE_ONLY=1 the library runs its bitstream stages only
        else if (!strcmp(argv[i], "--no-hw-check")) hw_check = false;
        else if (!in_path) in_path = argv[i];
        else if (!out_path) out_path = argv[i];
    }
    if (!in_path) { fprintf(stderr, "usage: %s in.h264 [out.yuv] [--codec 0|1] [--fmt 0|1] [--loops N] [--chunk BYTES]\n", argv[0]); return 2; }
    if (hw_check && !jm_nvdec_is_hw_support()) { fprintf(stderr, "test_amd_dec: no HIP device (there is no CPU fallback)\n"); return 3; }
    FILE *ifile = fopen(in_path, "rb");
    if (!ifile) { perror(in_path); return 2; }
    FILE *ofile = out_path ? fopen(out_path, "wb") : nullptr;
    if (out_path && !ofile) { perror(out_path); return 2; }
    if (chunk < 64) chunk = 64;

    handle_nvdec dec = jm_nvdec_create_handle();
    if (jm_nvdec_init(codec, fmt, nullptr, 0, dec) != 0) { fprintf(stderr, "test_amd_dec: jm_nvdec_init failed\n"); return 4; }

    std::vector<unsigned char> in_buf((size_t)chunk), out_buf(64u << 20);
    long buf_len = 0, nalu_count = 0; unsigned long frame_count = 0;
    int got_frame = 0, loops_left = loops;
    bool is_eof = false;
    auto fetch = [&] {
        int yuv_len = (int)out_buf.size();
        jm_nvdec_output_frame(out_buf.data(), &yuv_len, dec);
        if (yuv_len > 0) { frame_count++; if (ofile) fwrite(out_buf.data(), 1, (size_t)yuv_len, ofile); }
    };
    const auto t0 = std::chrono::steady_clock::now();
    while (!jm_nvdec_is_exit(dec)) {
        if (!is_eof) {
            long end = buf_len > 3 ? next_start_code(in_buf.data(), buf_len) : -1;
            if (end < 0) {                                                         // need more data
                if ((size_t)buf_len == in_buf.size()) in_buf.resize(in_buf.size() * 2);      // a NAL unit larger than the window
                size_t n = fread(in_buf.data() + buf_len, 1, in_buf.size() - (size_t)buf_len, ifile);
                if (n == 0 && --loops_left > 0) { rewind(ifile); n = fread(in_buf.data() + buf_len, 1, in_buf.size() - (size_t)buf_len, ifile); }
                buf_len += (long)n;
                if (n > 0) continue;
                is_eof = true; end = buf_len;                                      // last NAL unit of the file
            }
            if (end > 0) {
                nalu_count++;
                if (jm_nvdec_decode_frame(in_buf.data(), (int)end, &got_frame, dec) != 0) { fprintf(stderr, "test_amd_dec: jm_nvdec_decode_frame failed\n");
                    return 5; }
                if (got_frame == 1) fetch();
                memmove(in_buf.data(), in_buf.data() + end, (size_t)(buf_len - end));
                buf_len -= end;
            }
        } else {                                                                   // decode cached frames (test_nv_dec.cpp:234-247)
            nalu_count++;
            if (jm_nvdec_decode_frame(nullptr, 0, &got_frame, dec) != 0) { fprintf(stderr, "test_amd_dec: drain failed\n"); return 5; }
            if (got_frame == 1) fetch();
        }
    }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("%s", jm_nvdec_show_dec_info(dec));
    printf("------nalu count = %ld\n", nalu_count);
    int w = 0, h = 0; jm_nvdec_stream_info(&w, &h, dec);
    printf("------frames fetched = %lu (%dx%d), wall %.1f ms, %.1f frames/s\n", frame_count, w, h, wall_ms, wall_ms > 0 ? frame_count * 1000.0 / wall_ms : 0.0);
    jm_nvdec_deinit(dec);
    fclose(ifile);
    if (ofile) fclose(ofile);
    return 0;
}
