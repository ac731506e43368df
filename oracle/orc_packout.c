/*
 * oracle/orc_packout.c -- CPU ORACLE (test infrastructure only).
 * Byte-for-byte restatement of jm_nvdec_output_frame
 * (/root/reference/nv_dec/nv_dec.cpp:750-828): strip the pitch from a
 * pitch-linear NV12 surface and either copy the interleaved chroma rows
 * (out_fmt 0, :782-796) or de-interleave into a U plane followed by a V plane
 * (out_fmt 1, labelled "YV12" but written in I420 order, :798-820).
 * Return value and *out_len follow :757-758, :773-774, :824-827.
 */
#include "orc_h264.h"
#include <string.h>

int orc_packout(const uint8_t *src, int pitch, int width, int height, int out_fmt,
                uint8_t *dst, int *out_len) {
    if (!src) return -1;                                   /* :768-771 */
    if (*out_len < width * height * 3 / 2) return -2;      /* :773-774 */
    *out_len = 0;
    const uint8_t *py = src, *puv = src + (size_t)pitch * height;
    int xy = width * height;
    for (int y = 0; y < height; y++) { memcpy(dst + (size_t)y * width, py, (size_t)width); py += pitch; }
    if (out_fmt == 0) {
        int h2 = height >> 1;
        for (int y = 0; y < h2; y++) { memcpy(dst + xy + (size_t)y * width, puv, (size_t)width); puv += pitch; }
    } else {
        int w2 = width >> 1, h2 = height >> 1, uv = w2 * h2;
        for (int y = 0; y < h2; y++) {
            for (int x = 0; x < w2; x++) {
                dst[xy + y * w2 + x] = puv[x * 2];
                dst[xy + uv + y * w2 + x] = puv[x * 2 + 1];
            }
            puv += pitch;
        }
    }
    *out_len = width * height * 3 / 2;
    return *out_len;
}
