/*
 * jpeg_decode.c -- baseline JPEG decoder for image textures (product code, host).
 *
 * The reference loads images/earthmap.jpg through stb_image v2.26 (03_FullTracer/texture.c:63-70 -> stbi_load(..., 3)).  stb is
 * third-party code that lives only inside the reference tree and is not vendored here; this is a decoder of the published
 * format (ITU-T T.81 baseline sequential DCT, Huffman coding, 8-bit samples; JFIF YCbCr) whose three lossy stages are
 * restated from the algorithms stb_image documents, operation for operation, so that the TEXELS ARE BIT-IDENTICAL to the
 * reference's (a nearest-texel lookup, texture.c:136-152, shows every level of difference):
 *   - inverse DCT: the IJG "islow" integer IDCT (jidctint) with 12-bit constants, column pass kept at 2 extra bits, a DC-only
 *     column shortcut, row pass rounded at 17 bits with the +128 level shift folded in (stb_image.h:2352-2450);
 *   - chroma upsampling, row by row: 2x horizontally (3/4, 1/4), 2x vertically (3/4, 1/4), both (9/16 .. 1/16, rounded once),
 *     nearest for anything else; the far row alternates as in stb's ystep logic (stb_image.h:3364-3426, 3794-3836);
 *   - YCbCr -> RGB in 20-bit fixed point with 12-bit coefficients, the Cb term of G truncated to 16 bits before the sum
 *     (stb_image.h:3556-3582); component ids "RGB" (or Adobe transform 0 without JFIF) mean no conversion.
 *   - 1 or 3 components, sampling factors 1 or 2, restart intervals.
 *   - progressive files (SOF2: spectral selection and successive approximation, T.81 annex G): the scans fill 16-bit coefficient
 *     planes -- DC first / refinement, AC first / refinement with end-of-band runs -- which are dequantised (the product wraps to
 *     16 bits as in stb) and transformed once the last scan is in; from there on the path is the baseline one.
 * Arithmetic-coded, lossless, 12-bit and CMYK files and baseline files of more than one scan are refused (return 0), like any
 * file that is cut short.
 * tests/test_host_api.py: equal to the reference's own stbi_load output on every sample of images/earthmap.jpg and of
 * 4:2:0 / 4:2:2 / 4:4:0 / grey / restart-interval files (tests/golden/jpeg_cases.npz, made by the compiled reference).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "scene_internal.h"

typedef struct { unsigned char bits[17]; unsigned char vals[256]; int mincode[17], maxcode[18], valptr[17]; int present; } huff_table;
typedef struct { int id, h, v, tq, td, ta; int w, hgt, stride; unsigned char *plane; int pred; short *coef; int blocks_w, blocks_h; } component;
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

/* One 8-point pass of the IJG islow IDCT on s[0], s[st], ..., s[7 st]; constants = round(c * 4096).  Outputs the even part
 * x[0..3] and the odd part t[0..3] scaled by 4096: the caller adds its rounding bias to x and forms x[k] +- t[3 - k]. */
/* (All sums and products wrap modulo 2^32 -- unsigned arithmetic, read back as int -- so that a corrupt file's out-of-range
 * coefficients give the values two's-complement hardware gives the reference's int arithmetic instead of undefined behaviour;
 * for coefficients a valid file can hold nothing overflows.) */
typedef unsigned int u32;
static void idct_1d(const int *s, int st, int x[4], int t[4]) {
    u32 p2 = (u32)s[2 * st], p3 = (u32)s[6 * st];
    u32 p1 = (p2 + p3) * 2217u;
    const u32 e2 = p1 + p3 * (u32)-7567, e3 = p1 + p2 * 3135u;
    p2 = (u32)s[0]; p3 = (u32)s[4 * st];
    const u32 e0 = (p2 + p3) * 4096u, e1 = (p2 - p3) * 4096u;
    x[0] = (int)(e0 + e3); x[3] = (int)(e0 - e3); x[1] = (int)(e1 + e2); x[2] = (int)(e1 - e2);
    u32 t0 = (u32)s[7 * st], t1 = (u32)s[5 * st], t2 = (u32)s[3 * st], t3 = (u32)s[st];
    p3 = t0 + t2; u32 p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
    const u32 p5 = (p3 + p4) * 4816u;
    t0 *= 1223u; t1 *= 8410u; t2 *= 12586u; t3 *= 6149u;
    p1 = p5 + p1 * (u32)-3685; p2 = p5 + p2 * (u32)-10497; p3 *= (u32)-8034; p4 *= (u32)-1597;
    t[3] = (int)(t3 + p1 + p4); t[2] = (int)(t2 + p2 + p3); t[1] = (int)(t1 + p2 + p4); t[0] = (int)(t0 + p1 + p3);
}
static unsigned char clamp8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static int wsum(int a, int b) { return (int)((u32)a + (u32)b); }
static int wdif(int a, int b) { return (int)((u32)a - (u32)b); }
static void idct_block(const short *coef, unsigned char *out, int stride) {
    int val[64];
    for (int c = 0; c < 8; c++) {                                      /* columns; 2 extra bits kept */
        const short *d = coef + c; int *v = val + c;
        if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
            const int dc = d[0] * 4;
            for (int k = 0; k < 8; k++) v[8 * k] = dc;
            continue;
        }
        int in[8], x[4], t[4];
        for (int k = 0; k < 8; k++) in[k] = d[8 * k];
        idct_1d(in, 1, x, t);
        for (int k = 0; k < 4; k++) { x[k] = wsum(x[k], 512); v[8 * k] = wsum(x[k], t[3 - k]) >> 10; v[8 * (7 - k)] = wdif(x[k], t[3 - k]) >> 10; }
    }
    for (int r = 0; r < 8; r++) {                                      /* rows: 12 + 2 + 3 bits to drop, level shift +128 folded in */
        int x[4], t[4];
        idct_1d(val + 8 * r, 1, x, t);
        unsigned char *o = out + (size_t)r * stride;
        for (int k = 0; k < 4; k++) { x[k] = wsum(x[k], 65536 + (128 << 17)); o[k] = clamp8(wsum(x[k], t[3 - k]) >> 17); o[7 - k] = clamp8(wdif(x[k], t[3 - k]) >> 17); }
    }
}

/* One output row of a component at full width: `near` is the component row the output row lies in, `far` its neighbour on the
 * side the output row leans to (the same row at the image's top and bottom edge).  w = samples of the component in a row. */
static const unsigned char *upsample_row(unsigned char *out, const unsigned char *near, const unsigned char *far, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) return near;
    if (hs == 1 && vs == 2) { for (int i = 0; i < w; i++) out[i] = (unsigned char)((3 * near[i] + far[i] + 2) >> 2); return out; }
    if (hs == 2 && vs == 1) {
        if (w == 1) { out[0] = out[1] = near[0]; return out; }
        out[0] = near[0]; out[1] = (unsigned char)((near[0] * 3 + near[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; i++) { const int n = 3 * near[i] + 2; out[2 * i] = (unsigned char)((n + near[i - 1]) >> 2); out[2 * i + 1] = (unsigned char)((n + near[i + 1]) >> 2); }
        out[2 * i] = (unsigned char)((near[w - 2] * 3 + near[w - 1] + 2) >> 2); out[2 * i + 1] = near[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (unsigned char)((3 * near[0] + far[0] + 2) >> 2); return out; }
        int t1 = 3 * near[0] + far[0];
        out[0] = (unsigned char)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) { const int t0 = t1; t1 = 3 * near[i] + far[i]; out[2 * i - 1] = (unsigned char)((3 * t0 + t1 + 8) >> 4); out[2 * i] = (unsigned char)((3 * t1 + t0 + 8) >> 4); }
        out[2 * w - 1] = (unsigned char)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++) for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];   /* other factors: nearest */
    return out;
}
/* planes -> RGB8, row by row: upsample every component to full width, then convert (stb_image.h:3794-3850); NULL when out of memory */
static unsigned char *planes_to_rgb(component *comp, int ncomp, int W, int H, int hmax, int vmax, int is_rgb) {
    unsigned char *rgb = (unsigned char *)malloc((size_t)W * H * 3);
    if (!rgb) return NULL;
    unsigned char *line[3] = {NULL, NULL, NULL};
    const unsigned char *row0[3], *row1[3]; int ystep[3], ypos[3], hs[3], vs[3], wl[3];
    int alloc_ok = 1;
    for (int c = 0; c < ncomp; c++) {
        line[c] = (unsigned char *)malloc((size_t)W + 3 + 8);
        if (!line[c]) alloc_ok = 0;
        hs[c] = hmax / comp[c].h; vs[c] = vmax / comp[c].v; ystep[c] = vs[c] >> 1; ypos[c] = 0; wl[c] = (W + hs[c] - 1) / hs[c];
        row0[c] = row1[c] = comp[c].plane;
    }
    for (int y = 0; y < H && alloc_ok; y++) {
        const unsigned char *cur[3] = {NULL, NULL, NULL};
        for (int c = 0; c < ncomp; c++) {
            const int bot = ystep[c] >= (vs[c] >> 1);
            cur[c] = upsample_row(line[c], bot ? row1[c] : row0[c], bot ? row0[c] : row1[c], wl[c], hs[c], vs[c]);
            if (++ystep[c] >= vs[c]) { ystep[c] = 0; row0[c] = row1[c]; if (++ypos[c] < comp[c].hgt) row1[c] += comp[c].stride; }
        }
        unsigned char *o = rgb + (size_t)y * W * 3;
        for (int x = 0; x < W; x++, o += 3) {
            if (ncomp == 1) { o[0] = o[1] = o[2] = cur[0][x]; continue; }
            if (is_rgb) { o[0] = cur[0][x]; o[1] = cur[1][x]; o[2] = cur[2][x]; continue; }
            const int yf = (cur[0][x] << 20) + (1 << 19), cb = cur[1][x] - 128, cr = cur[2][x] - 128;
            /* JFIF: R = Y + 1.402 Cr, G = Y - 0.71414 Cr - 0.34414 Cb, B = Y + 1.772 Cb; coefficients round(c * 4096) << 8;
             * the Cb term of G loses its low 16 bits before it is added (stb's SIMD-compatible form) */
            o[0] = clamp8((yf + cr * 1470208) >> 20);
            o[1] = clamp8((int)(yf + cr * -748800 + (int)((unsigned)(cb * -360960) & 0xffff0000u)) >> 20);
            o[2] = clamp8((yf + cb * 1858048) >> 20);
        }
    }
    for (int c = 0; c < 3; c++) free(line[c]);
    if (!alloc_ok) { free(rgb); return NULL; }
    return rgb;
}

/* ---- progressive scans (T.81 annex G) ------------------------------------------------------------------------------------------ */
typedef struct { int ss, se, ah, al, eob_run; } prog_scan;
/* one block of a DC scan: the first pass leaves the prediction shifted left by Al, a refinement pass adds one bit */
static int prog_dc(bitreader *br, const huff_table *t, component *c, short *d, const prog_scan *ps) {
    if (ps->ah == 0) {
        const int n = huff_decode(br, t);
        if (n < 0 || n > 15) return 0;
        c->pred += n ? extend(get_bits(br, n), n) : 0;
        d[0] = (short)(c->pred * (1 << ps->al));
    } else if (next_bit(br)) d[0] = (short)(d[0] + (short)(1 << ps->al));
    return 1;
}
/* one block of an AC scan over the band ss..se */
static int prog_ac(bitreader *br, const huff_table *t, short *d, prog_scan *ps) {
    if (ps->ah == 0) {                                               /* first pass: values shifted left by Al, bands of zero blocks as EOB runs */
        if (ps->eob_run) { --ps->eob_run; return 1; }
        int k = ps->ss;
        do {
            const int rs = huff_decode(br, t);
            if (rs < 0) return 0;
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) { ps->eob_run = 1 << r; if (r) ps->eob_run += get_bits(br, r); --ps->eob_run; break; }
                k += 16;
            } else {
                k += r;
                if (k > 63) return 0;
                d[kZigzag[k++]] = (short)(extend(get_bits(br, s), s) * (1 << ps->al));
            }
        } while (k <= ps->se);
        return 1;
    }
    const short bit = (short)(1 << ps->al);                          /* refinement: one more bit for the coefficients that are set, new +-1 ones between them */
    if (ps->eob_run) {
        --ps->eob_run;
        for (int k = ps->ss; k <= ps->se; ++k) {
            short *p = d + kZigzag[k];
            if (*p != 0 && next_bit(br) && (*p & bit) == 0) *p = (short)(*p > 0 ? *p + bit : *p - bit);
        }
        return 1;
    }
    int k = ps->ss;
    do {
        const int rs = huff_decode(br, t);
        if (rs < 0) return 0;
        int s = rs & 15, r = rs >> 4;
        if (s == 0) {
            if (r < 15) { ps->eob_run = (1 << r) - 1; if (r) ps->eob_run += get_bits(br, r); r = 64; }   /* the rest of this band, then eob_run more blocks */
        } else {
            if (s != 1) return 0;
            s = next_bit(br) ? bit : -bit;
        }
        while (k <= ps->se) {
            short *p = d + kZigzag[k++];
            if (*p != 0) { if (next_bit(br) && (*p & bit) == 0) *p = (short)(*p > 0 ? *p + bit : *p - bit); }
            else { if (r == 0) { *p = (short)s; break; } --r; }
        }
    } while (k <= ps->se);
    return 1;
}
/* RSTn between two restart intervals: realign, reset predictions and the EOB run; 0 when the marker is not there */
static int take_restart(bitreader *br, component *comp, int ncomp, prog_scan *ps) {
    br->bitcnt = 0;
    if (!br->marker) { while (br->p + 1 < br->end && !(br->p[0] == 0xFF && br->p[1] >= 0xD0 && br->p[1] <= 0xD7)) br->p++; }
    if (br->p + 1 >= br->end || br->p[0] != 0xFF || br->p[1] < 0xD0 || br->p[1] > 0xD7) return 0;
    br->p += 2; br->marker = 0;
    for (int c = 0; c < ncomp; c++) comp[c].pred = 0;
    if (ps) ps->eob_run = 0;
    return 1;
}

/* Returns 1 and a malloc'ed RGB8 image (rows top to bottom) on success, 0 on any malformed or unsupported file. */
int rt_jpeg_decode(const unsigned char *data, size_t n, int *width, int *height, unsigned char **rgb_out) {
    if (!data || n < 4 || data[0] != 0xFF || data[1] != 0xD8) return 0;
    unsigned short qt[4][64]; int qt_present[4] = {0, 0, 0, 0};
    huff_table *dc = (huff_table *)calloc(4, sizeof(huff_table)), *ac = (huff_table *)calloc(4, sizeof(huff_table));
    component comp[3]; memset(comp, 0, sizeof comp);
    int ncomp = 0, W = 0, H = 0, hmax = 1, vmax = 1, restart = 0, ok = 0, have_frame = 0, jfif = 0, adobe_transform = -1, is_rgb = 0;
    int progressive = 0, scans = 0;
    size_t pos = 2;
    unsigned char *rgb = NULL;
    if (!dc || !ac) goto done;
    while (pos + 4 <= n) {
        if (data[pos] != 0xFF) goto done;
        int m = data[pos + 1];
        if (m == 0xFF) { pos++; continue; }                            /* fill byte */
        pos += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;                                           /* EOI: a progressive file is complete here; a baseline one had no scan */
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
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {               /* SOF0 / SOF1: sequential DCT, Huffman; SOF2: progressive */
            if (have_frame || sl < 6 || seg[0] != 8) goto done;
            progressive = m == 0xC2;
            H = (seg[1] << 8) | seg[2]; W = (seg[3] << 8) | seg[4]; ncomp = seg[5];
            if (W <= 0 || H <= 0 || (ncomp != 1 && ncomp != 3) || sl < (size_t)(6 + 3 * ncomp) || (long long)W * H > (1ll << 28)) goto done;
            for (int c = 0; c < ncomp; c++) {
                comp[c].id = seg[6 + 3 * c]; comp[c].h = seg[7 + 3 * c] >> 4; comp[c].v = seg[7 + 3 * c] & 15; comp[c].tq = seg[8 + 3 * c];
                if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) goto done;
                if (comp[c].h > hmax) hmax = comp[c].h;
                if (comp[c].v > vmax) vmax = comp[c].v;
            }
            have_frame = 1;
            if (progressive) {                                          /* coefficient planes, MCU-padded, zero to start with */
                const int mcus_x = (W + 8 * hmax - 1) / (8 * hmax), mcus_y = (H + 8 * vmax - 1) / (8 * vmax);
                for (int c = 0; c < ncomp; c++) {
                    comp[c].blocks_w = mcus_x * comp[c].h; comp[c].blocks_h = mcus_y * comp[c].v;
                    comp[c].stride = comp[c].blocks_w * 8; comp[c].w = (W * comp[c].h + hmax - 1) / hmax; comp[c].hgt = (H * comp[c].v + vmax - 1) / vmax;
                    comp[c].coef = (short *)calloc((size_t)comp[c].blocks_w * comp[c].blocks_h * 64, sizeof(short));
                    comp[c].plane = (unsigned char *)calloc((size_t)comp[c].stride * (size_t)(comp[c].blocks_h * 8), 1);
                    if (!comp[c].coef || !comp[c].plane) goto done;
                }
            }
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            goto done;                                                  /* lossless, differential, arithmetic: unsupported */
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
        } else if (m == 0xE0) {                                         /* APP0: "JFIF\0" */
            if (sl >= 5 && memcmp(seg, "JFIF", 5) == 0) jfif = 1;
        } else if (m == 0xEE) {                                         /* APP14: "Adobe\0", colour transform in its 12th byte */
            if (sl >= 12 && memcmp(seg, "Adobe", 6) == 0) adobe_transform = seg[11];
        } else if (m == 0xDD) {                                         /* DRI */
            if (sl < 2) goto done;
            restart = (seg[0] << 8) | seg[1];
        } else if (m == 0xDA && progressive) {                          /* SOS of a progressive file: one of many scans */
            if (!have_frame || sl < 1) goto done;
            const int ns = seg[0];
            if (ns < 1 || ns > ncomp || sl < (size_t)(1 + 2 * ns + 3)) goto done;
            int sc_comp[3];
            for (int q = 0; q < ns; q++) {
                int id = seg[1 + 2 * q], c;
                for (c = 0; c < ncomp && comp[c].id != id; c++) {}
                if (c == ncomp) goto done;
                comp[c].td = seg[2 + 2 * q] >> 4; comp[c].ta = seg[2 + 2 * q] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3) goto done;
                sc_comp[q] = c;
            }
            prog_scan ps; ps.ss = seg[1 + 2 * ns]; ps.se = seg[2 + 2 * ns]; ps.ah = seg[3 + 2 * ns] >> 4; ps.al = seg[3 + 2 * ns] & 15; ps.eob_run = 0;
            if (ps.ss > 63 || ps.se > 63 || ps.ss > ps.se || ps.ah > 13 || ps.al > 13) goto done;
            if ((ps.ss == 0) != (ps.se == 0)) goto done;                 /* a scan is DC only or AC only */
            if (ps.ss != 0 && ns != 1) goto done;                        /* AC scans hold one component */
            for (int q = 0; q < ns; q++) {
                const component *cc = comp + sc_comp[q];
                if (ps.ss == 0 ? (ps.ah == 0 && !dc[cc->td].present) : !ac[cc->ta].present) goto done;
            }
            bitreader br; br.p = data + pos + len; br.end = data + n; br.bitbuf = 0; br.bitcnt = 0; br.marker = 0;
            for (int c = 0; c < ncomp; c++) comp[c].pred = 0;
            int until_restart = restart;
            if (ns == 1) {                                               /* one component: its own blocks, row by row, no MCU padding */
                component *cc = comp + sc_comp[0];
                const int bw = (cc->w + 7) >> 3, bh = (cc->hgt + 7) >> 3;
                for (int j = 0; j < bh; j++)
                    for (int i = 0; i < bw; i++) {
                        if (restart && until_restart == 0) { if (!take_restart(&br, comp, ncomp, &ps)) goto done; until_restart = restart; }
                        short *d = cc->coef + ((size_t)j * cc->blocks_w + i) * 64;
                        if (!(ps.ss == 0 ? prog_dc(&br, dc + cc->td, cc, d, &ps) : prog_ac(&br, ac + cc->ta, d, &ps))) goto done;
                        if (restart) until_restart--;
                    }
            } else {                                                     /* interleaved: DC scans only */
                const int mcus_x = (W + 8 * hmax - 1) / (8 * hmax), mcus_y = (H + 8 * vmax - 1) / (8 * vmax);
                for (int my = 0; my < mcus_y; my++)
                    for (int mx = 0; mx < mcus_x; mx++) {
                        if (restart && until_restart == 0) { if (!take_restart(&br, comp, ncomp, &ps)) goto done; until_restart = restart; }
                        for (int q = 0; q < ns; q++) {
                            component *cc = comp + sc_comp[q];
                            for (int by = 0; by < cc->v; by++)
                                for (int bx = 0; bx < cc->h; bx++)
                                    if (!prog_dc(&br, dc + cc->td, cc, cc->coef + ((size_t)(my * cc->v + by) * cc->blocks_w + (size_t)(mx * cc->h + bx)) * 64, &ps)) goto done;
                        }
                        if (restart) until_restart--;
                    }
            }
            ++scans;
            /* on to the marker that ends the entropy-coded data */
            if (!br.marker) { while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] != 0 && !(br.p[1] >= 0xD0 && br.p[1] <= 0xD7) && br.p[1] != 0xFF)) br.p++; }
            pos = (size_t)(br.p - data);
            continue;
        } else if (m == 0xDA) {                                         /* SOS: the one scan of a baseline file */
            if (!have_frame || sl < 1 || seg[0] != ncomp || sl < (size_t)(1 + 2 * ncomp + 3)) goto done;
            for (int s = 0; s < ncomp; s++) {
                int id = seg[1 + 2 * s], c;
                for (c = 0; c < ncomp && comp[c].id != id; c++) {}
                if (c == ncomp) goto done;
                comp[c].td = seg[2 + 2 * s] >> 4; comp[c].ta = seg[2 + 2 * s] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3 || !dc[comp[c].td].present || !ac[comp[c].ta].present || !qt_present[comp[c].tq]) goto done;
            }
            is_rgb = ncomp == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') || (adobe_transform == 0 && !jfif));   /* stb_image.h:3778 */
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
                                short coef[64]; memset(coef, 0, sizeof coef);                /* 16-bit like stb's: the product wraps there too */
                                int t = huff_decode(&br, dc + comp[c].td);
                                if (t < 0 || t > 11) goto done;
                                comp[c].pred += extend(get_bits(&br, t), t);
                                coef[0] = (short)(comp[c].pred * qt[comp[c].tq][0]);
                                for (int k = 1; k < 64;) {
                                    int rs = huff_decode(&br, ac + comp[c].ta);
                                    if (rs < 0) goto done;
                                    int r = rs >> 4, s = rs & 15;
                                    if (s == 0) { if (r != 15) break; k += 16; continue; }
                                    k += r;
                                    if (k > 63) goto done;
                                    coef[kZigzag[k]] = (short)(extend(get_bits(&br, s), s) * qt[comp[c].tq][kZigzag[k]]);
                                    k++;
                                }
                                unsigned char *dst = comp[c].plane + (size_t)((my * comp[c].v + by) * 8) * comp[c].stride + (size_t)(mx * comp[c].h + bx) * 8;
                                idct_block(coef, dst, comp[c].stride);
                            }
                    if (br.marker && !(br.marker >= 0xD0 && br.marker <= 0xD7) && !(my == mcus_y - 1 && mx == mcus_x - 1)) goto done;   /* cut short */
                    if (restart) until_restart--;
                }
            rgb = planes_to_rgb(comp, ncomp, W, H, hmax, vmax, is_rgb);
            if (!rgb) goto done;
            *width = W; *height = H; *rgb_out = rgb; rgb = NULL; ok = 1;
            goto done;
        }
        pos += len;
    }
    if (progressive && have_frame && scans > 0) {                      /* every scan is in: dequantise (16-bit wrap, like stb), transform, convert */
        for (int c = 0; c < ncomp; c++) {
            if (!qt_present[comp[c].tq]) goto done;
            const int bw = (comp[c].w + 7) >> 3, bh = (comp[c].hgt + 7) >> 3;
            for (int j = 0; j < bh; j++)
                for (int i = 0; i < bw; i++) {
                    short *d = comp[c].coef + ((size_t)j * comp[c].blocks_w + i) * 64;
                    for (int k = 0; k < 64; k++) d[k] = (short)(d[k] * qt[comp[c].tq][k]);
                    idct_block(d, comp[c].plane + (size_t)(j * 8) * comp[c].stride + (size_t)i * 8, comp[c].stride);
                }
        }
        is_rgb = ncomp == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') || (adobe_transform == 0 && !jfif));   /* stb_image.h:3778 */
        rgb = planes_to_rgb(comp, ncomp, W, H, hmax, vmax, is_rgb);
        if (!rgb) goto done;
        *width = W; *height = H; *rgb_out = rgb; rgb = NULL; ok = 1;
    }
done:
    for (int c = 0; c < 3; c++) { free(comp[c].plane); free(comp[c].coef); }
    free(dc); free(ac); free(rgb);
    return ok;
}
