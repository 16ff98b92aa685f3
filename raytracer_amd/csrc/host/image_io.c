/*
 * image_io.c -- PNG / PPM output for the host (product code).
 * The reference writes its PNG through stb_image_write (03_FullTracer/main.c:170-171).  stb is third-party code
 * that only exists inside the reference tree, so it is not vendored here; this encoder produces a standard
 * 8-bit RGB PNG (filter 0, zlib deflate) with identical decoded pixels and the same vertical-flip convention.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "raytracer.h"

static void put_be32(unsigned char *p, unsigned long v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; }
static int write_chunk(FILE *f, const char *tag, const unsigned char *data, size_t n) {
    unsigned char hdr[8]; put_be32(hdr, (unsigned long)n); memcpy(hdr + 4, tag, 4);
    unsigned long crc = crc32(0L, hdr + 4, 4);
    if (n) crc = crc32(crc, data, (uInt)n);
    unsigned char tail[4]; put_be32(tail, crc);
    return fwrite(hdr, 1, 8, f) == 8 && (n == 0 || fwrite(data, 1, n, f) == n) && fwrite(tail, 1, 4, f) == 4;
}
int rt_write_png(const char *path, int w, int h, const unsigned char *rgb, int flip_vertically) {
    if (!path || !rgb || w <= 0 || h <= 0) return 0;
    const size_t stride = (size_t)w * 3, raw_n = (stride + 1) * (size_t)h;
    unsigned char *raw = (unsigned char *)malloc(raw_n);
    if (!raw) return 0;
    for (int y = 0; y < h; y++) {
        const unsigned char *src = rgb + stride * (size_t)(flip_vertically ? h - 1 - y : y);
        raw[(stride + 1) * (size_t)y] = 0;
        memcpy(raw + (stride + 1) * (size_t)y + 1, src, stride);
    }
    uLongf zn = compressBound((uLong)raw_n);
    unsigned char *z = (unsigned char *)malloc(zn);
    int ok = z && compress2(z, &zn, raw, (uLong)raw_n, 6) == Z_OK;
    free(raw);
    FILE *f = ok ? fopen(path, "wb") : NULL;
    if (!f) { free(z); return 0; }
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    unsigned char ihdr[13]; put_be32(ihdr, (unsigned long)w); put_be32(ihdr + 4, (unsigned long)h);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    ok = fwrite(sig, 1, 8, f) == 8 && write_chunk(f, "IHDR", ihdr, 13) && write_chunk(f, "IDAT", z, zn) && write_chunk(f, "IEND", NULL, 0);
    free(z);
    return fclose(f) == 0 && ok;
}
int rt_write_ppm_p3(const char *path, int w, int h, const unsigned char *rgb, int flip_vertically) { /* 01 main.c:168,30-35 */
    FILE *f = fopen(path, "w");
    if (!f) return 0;
    fprintf(f, "P3\n%d %d\n255\n", w, h);
    for (int y = 0; y < h; y++) {
        const unsigned char *row = rgb + (size_t)w * 3 * (size_t)(flip_vertically ? h - 1 - y : y);
        for (int x = 0; x < w; x++) fprintf(f, "%d %d %d\n", row[3 * x], row[3 * x + 1], row[3 * x + 2]);
    }
    return fclose(f) == 0;
}
