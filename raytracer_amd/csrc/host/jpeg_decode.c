/*
 * jpeg_decode.c -- baseline JPEG decoder for image textures (product code, host).
 *
 * The reference loads images/earthmap.jpg through stb_image (03_FullTracer/texture.c:63-70).  stb is third-party code
 * that lives only inside the reference tree and is not vendored here; this is an independent implementation of the
 * published format (ITU-T T.81 baseline sequential DCT, Huffman coding, 8-bit samples; JFIF YCbCr):
 *   - 1 or 3 components, sampling factors 1 or 2, restart intervals;
 *   - exact separable inverse DCT in double precision, rounded once (stb uses a 12-bit integer IDCT: samples may differ
 *     by one level);
 *   - chroma upsampling by the usual triangle filter (3/4, 1/4) for 2x subsampled planes;
 *   - JFIF colour conversion in 20-bit fixed point.
 * Progressive (SOF2), arithmetic-coded, 12-bit and CMYK files are refused (return 0), like any file that is cut short.
 * tests/test_host_api.py pins it to pixels decoded by the reference's own loader (tests/golden/earthmap_decode.npz):
 * |difference| <= 3 of 255 on every sample, 99.8 % of the samples within 1.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "scene_internal.h"

typedef struct { unsigned char bits[17]; unsigned char vals[256]; int mincode[17], maxcode[18], valptr[17]; int present; } huff_table;
typedef struct { int id, h, v, tq, td, ta; int w, hgt, stride; unsigned char *plane; int pred; } component;
typedef struct {
    const unsigned char *p, *end;
    unsigned int bitbuf; int bitcnt; int marker; /* marker hit inside entropy data (0 = none) */
} bitreader;

static const unsigned char kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static void huff_build(huff_table *t) {
    int code = 0, k = 0;
    for (int len = 1; len <= 16; len++) {
        t->valptr[len] = k; t->mincode[len] = code;
        code += t->bits[len]; k += t->bits[len];
        t->maxcode[len] = t->bits[len] ? code - 1 : -1;
        code <<= 1;
    }
    t->maxcode[17] = 0x7fffffff; t->present = 1;
}
static int next_bit(bitreader *br) {
    if (br->bitcnt == 0) {
        if (br->marker) return 0;                                   /* past the data: feed zeros, caller checks marker */
        if (br->p >= br->end) { br->marker = 0xD9; return 0; }      /* file ends without EOI: same as meeting EOI here */
        unsigned int b = *br->p++;
        if (b == 0xFF) {
            unsigned int b2 = br->p < br->end ? *br->p : 0xD9;
            if (b2 == 0) br->p++;                                    /* stuffed byte */
            else { br->marker = (int)b2; br->p--; return 0; }        /* a real marker: leave it for the caller */
        }
        br->bitbuf = b; br->bitcnt = 8;
    }
    br->bitcnt--;
    return (int)((br->bitbuf >> br->bitcnt) & 1u);
}
static int get_bits(bitreader *br, int n) { int v = 0; while (n-- > 0) v = (v << 1) | next_bit(br); return v; }
static int huff_decode(bitreader *br, const huff_table *t) {
    int code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | next_bit(br);
        if (t->maxcode[len] >= 0 && code <= t->maxcode[len] && code >= t->mincode[len]) return t->vals[t->valptr[len] + code - t->mincode[len]];
    }
    return -1;
}
static int extend(int v, int n) { return n && v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

static double g_cos[8][8];
static void idct_init(void) {
    static int done = 0;
    if (done) return;
    for (int x = 0; x < 8; x++)
        for (int u = 0; u < 8; u++) g_cos[x][u] = (u == 0 ? sqrt(0.5) : 1.0) * 0.5 * cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
    done = 1;
}
static void idct_block(const int *coef, unsigned char *out, int stride) {
    double tmp[64];
    for (int v = 0; v < 8; v++)                                       /* rows of coefficients -> spatial x */
        for (int x = 0; x < 8; x++) { double s = 0; for (int u = 0; u < 8; u++) s += g_cos[x][u] * coef[8 * v + u]; tmp[8 * v + x] = s; }
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++) {
            double s = 0; for (int v = 0; v < 8; v++) s += g_cos[y][v] * tmp[8 * v + x];
            int q = (int)floor(s + 128.5);
            out[y * stride + x] = (unsigned char)(q < 0 ? 0 : (q > 255 ? 255 : q));
        }
}

/* sample of a component plane at full-resolution pixel (x, y), triangle filter for 2x subsampled directions */
static int plane_sample(const component *c, int hmax, int vmax, int x, int y) {
    const int sx = hmax / c->h, sy = vmax / c->v;
    if (sx == 1 && sy == 1) return c->plane[y * c->stride + x];
    int x0 = x, x1 = x, wx = 0, y0 = y, y1 = y, wy = 0;                /* weights in quarters for the second tap */
    if (sx == 2) { x0 = x >> 1; x1 = (x & 1) ? x0 + 1 : x0 - 1; wx = 1; } else if (sx > 2) x0 = x1 = x / sx;
    if (sy == 2) { y0 = y >> 1; y1 = (y & 1) ? y0 + 1 : y0 - 1; wy = 1; } else if (sy > 2) y0 = y1 = y / sy;
    x1 = x1 < 0 ? 0 : (x1 >= c->w ? c->w - 1 : x1);
    y1 = y1 < 0 ? 0 : (y1 >= c->hgt ? c->hgt - 1 : y1);
    const unsigned char *p = c->plane;
    const int a = p[y0 * c->stride + x0], b = p[y0 * c->stride + x1], d = p[y1 * c->stride + x0], e = p[y1 * c->stride + x1];
    const int near_w = wx ? 3 : 4, far_w = wx ? 1 : 0, near_h = wy ? 3 : 4, far_h = wy ? 1 : 0;
    const int top = near_w * a + far_w * b, bot = near_w * d + far_w * e;           /* x 4 */
    return (near_h * top + far_h * bot + 8) >> 4;
}
static unsigned char clamp8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* Returns 1 and a malloc'ed RGB8 image (rows top to bottom) on success, 0 on any malformed or unsupported file. */
int rt_jpeg_decode(const unsigned char *data, size_t n, int *width, int *height, unsigned char **rgb_out) {
    if (!data || n < 4 || data[0] != 0xFF || data[1] != 0xD8) return 0;
    idct_init();
    unsigned short qt[4][64]; int qt_present[4] = {0, 0, 0, 0};
    huff_table *dc = (huff_table *)calloc(4, sizeof(huff_table)), *ac = (huff_table *)calloc(4, sizeof(huff_table));
    component comp[3]; memset(comp, 0, sizeof comp);
    int ncomp = 0, W = 0, H = 0, hmax = 1, vmax = 1, restart = 0, ok = 0, have_frame = 0;
    size_t pos = 2;
    unsigned char *rgb = NULL;
    if (!dc || !ac) goto done;
    while (pos + 4 <= n) {
        if (data[pos] != 0xFF) goto done;
        int m = data[pos + 1];
        if (m == 0xFF) { pos++; continue; }                            /* fill byte */
        pos += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) goto done;                                       /* EOI before any scan */
        if (pos + 2 > n) goto done;
        size_t len = ((size_t)data[pos] << 8) | data[pos + 1];
        if (len < 2 || pos + len > n) goto done;
        const unsigned char *seg = data + pos + 2; size_t sl = len - 2;
        if (m == 0xDB) {                                                /* DQT */
            size_t i = 0;
            while (i < sl) {
                int pq = seg[i] >> 4, tq = seg[i] & 15; i++;
                if (tq > 3 || pq > 1 || i + (size_t)(pq ? 128 : 64) > sl) goto done;
                for (int k = 0; k < 64; k++) { qt[tq][kZigzag[k]] = pq ? (unsigned short)((seg[i] << 8) | seg[i + 1]) : seg[i]; i += pq ? 2 : 1; }
                qt_present[tq] = 1;
            }
        } else if (m == 0xC0 || m == 0xC1) {                            /* SOF0 / SOF1: sequential DCT, Huffman */
            if (sl < 6 || seg[0] != 8) goto done;
            H = (seg[1] << 8) | seg[2]; W = (seg[3] << 8) | seg[4]; ncomp = seg[5];
            if (W <= 0 || H <= 0 || (ncomp != 1 && ncomp != 3) || sl < (size_t)(6 + 3 * ncomp) || (long long)W * H > (1ll << 28)) goto done;
            for (int c = 0; c < ncomp; c++) {
                comp[c].id = seg[6 + 3 * c]; comp[c].h = seg[7 + 3 * c] >> 4; comp[c].v = seg[7 + 3 * c] & 15; comp[c].tq = seg[8 + 3 * c];
                if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) goto done;
                if (comp[c].h > hmax) hmax = comp[c].h;
                if (comp[c].v > vmax) vmax = comp[c].v;
            }
            have_frame = 1;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            goto done;                                                  /* progressive, lossless, arithmetic: unsupported */
        } else if (m == 0xC4) {                                         /* DHT */
            size_t i = 0;
            while (i < sl) {
                if (i + 17 > sl) goto done;
                int tc = seg[i] >> 4, th = seg[i] & 15; i++;
                if (tc > 1 || th > 3) goto done;
                huff_table *t = (tc ? ac : dc) + th;
                int total = 0; t->bits[0] = 0;
                for (int k = 1; k <= 16; k++) { t->bits[k] = seg[i++]; total += t->bits[k]; }
                if (total > 256 || i + (size_t)total > sl) goto done;
                memcpy(t->vals, seg + i, (size_t)total); i += (size_t)total;
                huff_build(t);
            }
        } else if (m == 0xDD) {                                         /* DRI */
            if (sl < 2) goto done;
            restart = (seg[0] << 8) | seg[1];
        } else if (m == 0xDA) {                                         /* SOS: the one scan of a baseline file */
            if (!have_frame || sl < 1 || seg[0] != ncomp || sl < (size_t)(1 + 2 * ncomp + 3)) goto done;
            for (int s = 0; s < ncomp; s++) {
                int id = seg[1 + 2 * s], c;
                for (c = 0; c < ncomp && comp[c].id != id; c++) {}
                if (c == ncomp) goto done;
                comp[c].td = seg[2 + 2 * s] >> 4; comp[c].ta = seg[2 + 2 * s] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3 || !dc[comp[c].td].present || !ac[comp[c].ta].present || !qt_present[comp[c].tq]) goto done;
            }
            const int mcu_w = 8 * hmax, mcu_h = 8 * vmax, mcus_x = (W + mcu_w - 1) / mcu_w, mcus_y = (H + mcu_h - 1) / mcu_h;
            for (int c = 0; c < ncomp; c++) {
                comp[c].stride = mcus_x * 8 * comp[c].h; comp[c].w = (W * comp[c].h + hmax - 1) / hmax; comp[c].hgt = (H * comp[c].v + vmax - 1) / vmax;
                comp[c].plane = (unsigned char *)calloc((size_t)comp[c].stride * (size_t)(mcus_y * 8 * comp[c].v), 1);
                if (!comp[c].plane) goto done;
                comp[c].pred = 0;
            }
            bitreader br; br.p = data + pos + len; br.end = data + n; br.bitbuf = 0; br.bitcnt = 0; br.marker = 0;
            int until_restart = restart;
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    if (restart && until_restart == 0) {                 /* expect RSTn: realign, reset predictors */
                        br.bitcnt = 0;
                        if (!br.marker) { while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) br.p++; }
                        if (br.p + 1 >= br.end || br.p[0] != 0xFF || br.p[1] < 0xD0 || br.p[1] > 0xD7) goto done;
                        br.p += 2; br.marker = 0;
                        for (int c = 0; c < ncomp; c++) comp[c].pred = 0;
                        until_restart = restart;
                    }
                    for (int c = 0; c < ncomp; c++)
                        for (int by = 0; by < comp[c].v; by++)
                            for (int bx = 0; bx < comp[c].h; bx++) {
                                int coef[64]; memset(coef, 0, sizeof coef);
                                int t = huff_decode(&br, dc + comp[c].td);
                                if (t < 0 || t > 11) goto done;
                                comp[c].pred += extend(get_bits(&br, t), t);
                                coef[0] = comp[c].pred * qt[comp[c].tq][0];
                                for (int k = 1; k < 64;) {
                                    int rs = huff_decode(&br, ac + comp[c].ta);
                                    if (rs < 0) goto done;
                                    int r = rs >> 4, s = rs & 15;
                                    if (s == 0) { if (r != 15) break; k += 16; continue; }
                                    k += r;
                                    if (k > 63) goto done;
                                    coef[kZigzag[k]] = extend(get_bits(&br, s), s) * qt[comp[c].tq][kZigzag[k]];
                                    k++;
                                }
                                unsigned char *dst = comp[c].plane + (size_t)((my * comp[c].v + by) * 8) * comp[c].stride + (size_t)(mx * comp[c].h + bx) * 8;
                                idct_block(coef, dst, comp[c].stride);
                            }
                    if (br.marker && !(br.marker >= 0xD0 && br.marker <= 0xD7) && !(my == mcus_y - 1 && mx == mcus_x - 1)) goto done;   /* cut short */
                    if (restart) until_restart--;
                }
            rgb = (unsigned char *)malloc((size_t)W * H * 3);
            if (!rgb) goto done;
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    unsigned char *o = rgb + ((size_t)y * W + x) * 3;
                    const int Y = plane_sample(&comp[0], hmax, vmax, x, y);
                    if (ncomp == 1) { o[0] = o[1] = o[2] = (unsigned char)Y; continue; }
                    const int cb = plane_sample(&comp[1], hmax, vmax, x, y) - 128, cr = plane_sample(&comp[2], hmax, vmax, x, y) - 128;
                    const int yf = (Y << 20) + (1 << 19);               /* JFIF: R = Y + 1.402 Cr, G = Y - 0.34414 Cb - 0.71414 Cr, B = Y + 1.772 Cb */
                    o[0] = clamp8((yf + cr * 1470208) >> 20);           /* round(1.402   * 2^20) at 12-bit precision: 5743 << 8 */
                    o[1] = clamp8((yf - cr * 748800 - cb * 360960) >> 20);   /* 2925 << 8, 1410 << 8 */
                    o[2] = clamp8((yf + cb * 1858048) >> 20);           /* 7258 << 8 */
                }
            *width = W; *height = H; *rgb_out = rgb; rgb = NULL; ok = 1;
            goto done;
        }
        pos += len;
    }
done:
    for (int c = 0; c < 3; c++) free(comp[c].plane);
    free(dc); free(ac); free(rgb);
    return ok;
}
