/* oracle/orc_main.c -- CLI around the CPU oracle (test infrastructure only).
 * usage: orc_decode in.h264 [out.yuv] [fmt: 1=I420 (default), 0=NV12]
 * Prints frame count, display size and an FNV-1a checksum of the output. */
#include "orc_h264.h"
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s in.h264 [out.yuv] [fmt]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc(n);
    if (fread(buf, 1, n, f) != (size_t)n) return 1;
    fclose(f);
    int fmt = argc > 3 ? atoi(argv[3]) : 1, w = 0, h = 0;
    uint8_t *out = NULL; size_t out_len = 0;
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    int frames = orc_decode_stream_to_buffer(buf, n, fmt, &out, &out_len, &w, &h);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (frames < 0) { fprintf(stderr, "decode failed\n"); return 1; }
    uint64_t hash = 1469598103934665603ull;
    for (size_t i = 0; i < out_len; i++) { hash ^= out[i]; hash *= 1099511628211ull; }
    double sec = (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
    printf("frames=%d size=%dx%d bytes=%zu fnv1a=%016llx sec=%.3f fps=%.1f\n", frames, w, h, out_len,
           (unsigned long long)hash, sec, frames / (sec > 0 ? sec : 1));
    if (argc > 2 && argv[2][0] != '-') { FILE *o = fopen(argv[2], "wb"); fwrite(out, 1, out_len, o); fclose(o); }
    orc_free(out); free(buf);
    return 0;
}
