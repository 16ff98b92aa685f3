/* Mutation fuzz of the image-texture decoders -- rt_jpeg_decode (baseline and progressive), rt_png_decode, rt_bmp_decode, picked by the
 * first bytes of the seed file (development aid and tests/test_host_api.py::test_jpeg_decoder_under_sanitizers): byte flips in the
 * header and anywhere, truncations, stray markers, runs of 0xFF; every input sits in a heap block of exactly its size so that ASan sees
 * any over-read.  gcc -O1 -g -fsanitize=address,undefined -static-libasan jpeg_fuzz.c ../../raytracer_amd/csrc/host/{jpeg,image}_decode.c -lm */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int rt_jpeg_decode(const unsigned char *data, size_t n, int *width, int *height, unsigned char **rgb_out);
int rt_png_decode(const unsigned char *data, size_t n, int *width, int *height, unsigned char **rgb_out);
int rt_bmp_decode(const unsigned char *data, size_t n, int *width, int *height, unsigned char **rgb_out);
static unsigned long long s = 88172645463325252ull;
static unsigned rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *orig = malloc(n); if (fread(orig, 1, n, f) != (size_t)n) return 2; fclose(f);
    int iters = atoi(argv[2]), ok = 0, bad = 0;
    int (*decode)(const unsigned char *, size_t, int *, int *, unsigned char **) = n > 2 && orig[0] == 137 ? rt_png_decode : (n > 2 && orig[0] == 'B' ? rt_bmp_decode : rt_jpeg_decode);
    const long head = n < 700 ? n : 700;
    for (int it = 0; it < iters; it++) {
        long m = n;
        unsigned char *buf = malloc(n); memcpy(buf, orig, n);
        int kind = rnd() % 4;
        if (kind == 0) { int k = 1 + rnd() % 8; for (int j = 0; j < k; j++) buf[rnd() % (rnd() % 2 ? head : n)] = (unsigned char)rnd(); }   /* header-heavy byte flips */
        else if (kind == 1) m = 2 + rnd() % n;                                                /* truncation */
        else if (kind == 2) { long p = rnd() % n; buf[p] = 0xff; if (p + 1 < n) buf[p + 1] = 0xc0 + rnd() % 32; }   /* stray marker */
        else { long p = rnd() % head; int k = rnd() % 6; for (int j = 0; j < k && p + j < n; j++) buf[p + j] = 0xff; }
        unsigned char *exact = malloc(m); memcpy(exact, buf, m);                              /* exact-size block: ASan sees any overread */
        int w = 0, h = 0; unsigned char *rgb = NULL;
        int rc = decode(exact, (size_t)m, &w, &h, &rgb);
        if (rc) { ok++; if (w <= 0 || h <= 0 || !rgb) { printf("inconsistent success\n"); return 1; } volatile unsigned char t = rgb[(size_t)w * h * 3 - 1]; (void)t; free(rgb); } else bad++;
        free(exact); free(buf);
    }
    printf("decoded %d, refused %d\n", ok, bad);
    free(orig);
    return 0;
}
