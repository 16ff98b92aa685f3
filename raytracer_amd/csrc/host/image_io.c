/*
 * image_io.c -- PNG / PPM output for the host (product code).
 *
 * The reference writes its PNG through stb_image_write v1.15 (03_FullTracer/main.c:170-171: stbi_flip_vertically_on_write(1);
 * stbi_write_png(path, W, H, 3, pixels, W * 3)).  stb is third-party code that only exists inside the reference tree, so it is
 * not vendored; rt_write_png restates the published algorithm of that writer so that the FILE is byte-identical to the
 * reference's (tests/golden/png_reference.npz holds files written by the compiled reference):
 *   * per output row, the five PNG filters are tried in order 0..4 and the first one with the smallest sum of |signed byte|
 *     wins (external/stb_image_write.h:1105-1121); the top row has no row above it: there "up" is the raw row, "average"
 *     predicts from half the left byte and "paeth" from the left byte (:1062-1098);
 *   * one fixed-Huffman deflate block (:882-1010) over the filtered image: 3-byte hash into 16,384 buckets, a bucket keeps
 *     its last 8..16 positions (halved when it reaches 16), the longest match inside 32 KiB wins with ties to the newest
 *     position, and a match is dropped for a literal when the next byte starts a longer one ("lazy matching", level 8);
 *   * zlib header 78 5e, Adler-32 trailer; chunks IHDR, one IDAT, IEND.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raytracer.h"

/* ---- deflate bit stream: codes enter at the low end of a 32-bit window, whole bytes leave at once ---------------------- */
typedef struct { unsigned char *buf; size_t n, cap; unsigned window; int bits; int failed; } bit_sink;
static void sink_byte(bit_sink *s, unsigned char b) {
    if (s->failed) return;
    if (s->n == s->cap) {
        size_t cap = s->cap ? 2 * s->cap : 4096;
        unsigned char *nb = (unsigned char *)realloc(s->buf, cap);
        if (!nb) { s->failed = 1; return; }
        s->buf = nb; s->cap = cap;
    }
    s->buf[s->n++] = b;
}
static void sink_bits(bit_sink *s, unsigned value, int count) {          /* `count` bits of `value`, least significant first */
    s->window |= value << s->bits; s->bits += count;
    while (s->bits >= 8) { sink_byte(s, (unsigned char)(s->window & 0xffu)); s->window >>= 8; s->bits -= 8; }
}
static void sink_code(bit_sink *s, unsigned code, int count) {           /* a Huffman code goes out most significant bit first */
    unsigned r = 0;
    for (int k = 0; k < count; k++) r |= ((code >> k) & 1u) << (count - 1 - k);
    sink_bits(s, r, count);
}
/* the fixed literal/length code of RFC 1951 section 3.2.6 */
static void sink_symbol(bit_sink *s, int sym) {
    if (sym <= 143) sink_code(s, 0x30u + (unsigned)sym, 8);
    else if (sym <= 255) sink_code(s, 0x190u + (unsigned)(sym - 144), 9);
    else if (sym <= 279) sink_code(s, (unsigned)(sym - 256), 7);
    else sink_code(s, 0xc0u + (unsigned)(sym - 280), 8);
}

/* ---- the match finder's memory: per hash bucket the positions seen last, oldest first --------------------------------- */
enum { PNG_BUCKETS = 16384, PNG_LEVEL = 8, PNG_BUCKET_CAP = 2 * PNG_LEVEL, PNG_WINDOW = 32768, PNG_MAX_MATCH = 258 };
typedef struct { int count; int pos[PNG_BUCKET_CAP]; } bucket;
static unsigned hash3(const unsigned char *p) {
    unsigned h = (unsigned)p[0] + ((unsigned)p[1] << 8) + ((unsigned)p[2] << 16);
    h ^= h << 3; h += h >> 5; h ^= h << 4; h += h >> 17; h ^= h << 25; h += h >> 6;
    return h & (PNG_BUCKETS - 1);
}
static int run_length(const unsigned char *a, const unsigned char *b, int limit) {
    int n = 0;
    if (limit > PNG_MAX_MATCH) limit = PNG_MAX_MATCH;
    while (n < limit && a[n] == b[n]) n++;
    return n;
}
static const unsigned short kLenBase[30] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 259};
static const unsigned char kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const unsigned short kDistBase[31] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 32768};
static const unsigned char kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

/* zlib stream of data[0..n): malloc'ed, *out_n bytes; NULL when memory runs out */
static unsigned char *deflate_fixed(const unsigned char *data, int n, size_t *out_n) {
    bucket *table = (bucket *)calloc(PNG_BUCKETS, sizeof(bucket));
    if (!table) return NULL;
    bit_sink s; memset(&s, 0, sizeof s);
    sink_byte(&s, 0x78); sink_byte(&s, 0x5e);                              /* 32 KiB window, FLEVEL 1 */
    sink_bits(&s, 1, 1); sink_bits(&s, 1, 2);                              /* last block, fixed Huffman codes */
    int i = 0;
    while (i < n - 3) {
        bucket *b = &table[hash3(data + i)];
        int best = 3, from = -1;
        for (int k = 0; k < b->count; k++)
            if (b->pos[k] > i - PNG_WINDOW) {
                const int len = run_length(data + b->pos[k], data + i, n - i);
                if (len >= best) { best = len; from = b->pos[k]; }         /* ties go to the newer position */
            }
        if (b->count == PNG_BUCKET_CAP) { memmove(b->pos, b->pos + PNG_LEVEL, PNG_LEVEL * sizeof(int)); b->count = PNG_LEVEL; }
        b->pos[b->count++] = i;
        if (from >= 0) {                                                   /* would the next byte start a longer match? */
            const bucket *nb = &table[hash3(data + i + 1)];
            for (int k = 0; k < nb->count; k++)
                if (nb->pos[k] > i - (PNG_WINDOW - 1) && run_length(data + nb->pos[k], data + i + 1, n - i - 1) > best) { from = -1; break; }
        }
        if (from >= 0) {
            const int dist = i - from;
            int c = 0;
            while (best > kLenBase[c + 1] - 1) c++;
            sink_symbol(&s, 257 + c);
            if (kLenExtra[c]) sink_bits(&s, (unsigned)(best - kLenBase[c]), kLenExtra[c]);
            c = 0;
            while (dist > kDistBase[c + 1] - 1) c++;
            sink_code(&s, (unsigned)c, 5);
            if (kDistExtra[c]) sink_bits(&s, (unsigned)(dist - kDistBase[c]), kDistExtra[c]);
            i += best;
        } else sink_symbol(&s, data[i++]);
    }
    for (; i < n; i++) sink_symbol(&s, data[i]);
    sink_symbol(&s, 256);
    if (s.bits) sink_bits(&s, 0, 8 - s.bits);                              /* zero bits up to the byte boundary */
    free(table);
    unsigned long a = 1, c = 0;                                            /* Adler-32 of the input */
    for (int k = 0; k < n; k++) { a += data[k]; if (a >= 65521) a -= 65521; c = (c + a) % 65521; }
    sink_byte(&s, (unsigned char)(c >> 8)); sink_byte(&s, (unsigned char)c); sink_byte(&s, (unsigned char)(a >> 8)); sink_byte(&s, (unsigned char)a);
    if (s.failed) { free(s.buf); return NULL; }
    *out_n = s.n;
    return s.buf;
}

/* ---- PNG row filters ------------------------------------------------------------------------------------------------- */
static int paeth_predict(int left, int up, int up_left) {
    const int p = left + up - up_left, da = abs(p - left), db = abs(p - up), dc = abs(p - up_left);
    if (da <= db && da <= dc) return left;
    return db <= dc ? up : up_left;
}
/* filter `type` (0 none, 1 sub, 2 up, 3 average, 4 paeth) of row `cur`; prev = the output row above it, NULL for the top row,
 * whose missing neighbours the reference's writer does not treat as zeros throughout: its "up" is the raw row, its "average"
 * and "paeth" look at the left byte only (and at nothing for the first pixel) */
static void filter_row(const unsigned char *cur, const unsigned char *prev, int bytes, int bpp, int type, unsigned char *out) {
    for (int i = 0; i < bytes; i++) {
        const int left = i >= bpp ? cur[i - bpp] : 0, up = prev ? prev[i] : 0, up_left = (prev && i >= bpp) ? prev[i - bpp] : 0;
        int predict = 0;
        if (type == 1) predict = left;
        else if (type == 2) predict = up;
        else if (type == 3) predict = prev ? ((i >= bpp ? left + up : up) >> 1) : (left >> 1);
        else if (type == 4) predict = prev ? (i >= bpp ? paeth_predict(left, up, up_left) : up) : left;
        out[i] = (unsigned char)(cur[i] - predict);
    }
}

static void put_be32(unsigned char *p, unsigned long v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; }
static unsigned long crc32_png(const unsigned char *p, size_t n, unsigned long crc) {   /* CRC-32 (IEEE 802.3), bitwise */
    crc = ~crc & 0xfffffffful;
    for (size_t i = 0; i < n; i++) { crc ^= p[i]; for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ (0xedb88320ul & (0ul - (crc & 1ul))); }
    return ~crc & 0xfffffffful;
}
static int write_chunk(FILE *f, const char *tag, const unsigned char *data, size_t n) {
    unsigned char hdr[8]; put_be32(hdr, (unsigned long)n); memcpy(hdr + 4, tag, 4);
    unsigned long crc = crc32_png(hdr + 4, 4, 0);
    if (n) crc = crc32_png(data, n, crc);
    unsigned char tail[4]; put_be32(tail, crc);
    return fwrite(hdr, 1, 8, f) == 8 && (n == 0 || fwrite(data, 1, n, f) == n) && fwrite(tail, 1, 4, f) == 4;
}
int rt_write_png(const char *path, int w, int h, const unsigned char *rgb, int flip_vertically) {
    if (!path || !rgb || w <= 0 || h <= 0 || (long long)w * 3 + 1 > 0x7fffffffLL / h) return 0;
    const int bpp = 3, stride = w * bpp;
    const size_t raw_n = ((size_t)stride + 1) * (size_t)h;
    unsigned char *raw = (unsigned char *)malloc(raw_n), *trial = (unsigned char *)malloc((size_t)stride);
    if (!raw || !trial) { free(raw); free(trial); return 0; }
    for (int y = 0; y < h; y++) {
        const unsigned char *cur = rgb + (size_t)stride * (size_t)(flip_vertically ? h - 1 - y : y);
        const unsigned char *prev = y ? rgb + (size_t)stride * (size_t)(flip_vertically ? h - y : y - 1) : NULL;
        unsigned char *dst = raw + ((size_t)stride + 1) * (size_t)y;
        long best_cost = 0x7fffffffL; int best = 0;
        for (int type = 0; type < 5; type++) {
            filter_row(cur, prev, stride, bpp, type, trial);
            long cost = 0;
            for (int i = 0; i < stride; i++) cost += abs((int)(signed char)trial[i]);
            if (cost < best_cost) { best_cost = cost; best = type; memcpy(dst + 1, trial, (size_t)stride); }
        }
        dst[0] = (unsigned char)best;
    }
    free(trial);
    size_t zn = 0;
    unsigned char *z = deflate_fixed(raw, (int)raw_n, &zn);
    free(raw);
    FILE *f = z ? fopen(path, "wb") : NULL;
    if (!f) { free(z); return 0; }
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    unsigned char ihdr[13]; put_be32(ihdr, (unsigned long)w); put_be32(ihdr + 4, (unsigned long)h);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    int ok = fwrite(sig, 1, 8, f) == 8 && write_chunk(f, "IHDR", ihdr, 13) && write_chunk(f, "IDAT", z, zn) && write_chunk(f, "IEND", NULL, 0);
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
