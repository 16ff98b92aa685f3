/*
 * image_decode.c -- PNG, BMP and binary PGM / PPM readers for image textures (product code, host).
 *
 * The reference loads a texture through stb_image v2.26 -- stbi_load(filename, &w, &h, &n, 3), 03_FullTracer/texture.c:63-70 -- which
 * takes whatever format the file is in and hands back 8-bit RGB.  stb is third-party code that lives only inside the reference tree
 * and is not vendored here; these are decoders of the published formats (PNG: RFC 2083 + zlib RFC 1950 / deflate RFC 1951; BMP: the
 * Windows DIB headers) that resolve the formats' free choices the way stb_image does, so that the TEXELS ARE THE REFERENCE'S:
 *   PNG  all colour types and bit depths, Adam7 interlace; grey samples of 1 / 2 / 4 bits scale by 255 / 85 / 17, 16-bit samples keep
 *        their high byte, alpha (a channel or a tRNS chunk) is dropped without blending, palette entries are taken as they are;
 *        chunk CRCs and the zlib Adler-32 are not checked (stb checks neither);
 *   BMP  headers of 12 / 40 / 56 / 108 / 124 bytes; 1 / 4 / 8 bits with a palette, 24 bits, 16 / 32 bits through their bit masks
 *        (5-5-5 and 8-8-8-8 by default), an n-bit field expanded to 8 bits by repeating its bits; bottom-up unless the height is
 *        negative; RLE files are refused, as in stb;
 *   PNM  binary P5 (grey) and P6 (RGB), maxval 255.
 * A palette index beyond the PLTE chunk reads uninitialised memory in stb; here such a file is refused.  Files that are cut short
 * or inconsistent are refused (return 0) -- texture_image_init then prints the reference's message and returns NULL.
 * tests/test_host_api.py: equal to the compiled reference's own stbi_load output, sample for sample, on the files of
 * tests/golden/image_cases.npz (made by tests/golden/make_image_cases.py).
 */
#include <stdlib.h>
#include <string.h>

#include "scene_internal.h"

/* ---- inflate (RFC 1951) behind a zlib header (RFC 1950) ----------------------------------------------------------------------- */
typedef struct { const unsigned char *p, *end; unsigned int buf; int cnt; int bad; } zbits;
static unsigned int zget(zbits *z, int n) {              /* n <= 16 bits, least significant first */
    while (z->cnt < n) {
        if (z->p >= z->end) { z->bad = 1; return 0; }
        z->buf |= (unsigned int)*z->p++ << z->cnt; z->cnt += 8;
    }
    const unsigned int v = z->buf & ((1u << n) - 1u);
    z->buf >>= n; z->cnt -= n;
    return v;
}
typedef struct { unsigned short count[16], symbol[288]; } zhuff;
static int zhuff_build(zhuff *h, const unsigned char *len, int n) {
    unsigned short offs[16];
    memset(h->count, 0, sizeof h->count);
    for (int i = 0; i < n; i++) h->count[len[i]]++;
    h->count[0] = 0;
    int left = 1;
    for (int l = 1; l < 16; l++) { left = (left << 1) - h->count[l]; if (left < 0) return 0; }   /* over-subscribed */
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (unsigned short)(offs[l] + h->count[l]);
    for (int i = 0; i < n; i++) if (len[i]) h->symbol[offs[len[i]]++] = (unsigned short)i;
    return 1;
}
static int zhuff_decode(zbits *z, const zhuff *h) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
        code |= (int)zget(z, 1);
        if (z->bad) return -1;
        const int c = h->count[l];
        if (code - c < first) return h->symbol[index + (code - first)];
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}
/* data[0..n) = a zlib stream; returns a malloc'ed buffer of its first `want` bytes (what follows them is not looked at: stb only asks for
 * enough pixels), or NULL (cut short, malformed) */
static unsigned char *zlib_inflate(const unsigned char *data, size_t n, size_t want) {
    static const unsigned short len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const unsigned char len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const unsigned short dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const unsigned char dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const unsigned char order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    if (n < 2 || (data[0] & 15) != 8 || ((data[0] << 8) | data[1]) % 31 != 0 || (data[1] & 32)) return NULL;   /* deflate, valid check bits, no preset dictionary */
    unsigned char *out = (unsigned char *)malloc(want ? want : 1);
    zhuff *lit = (zhuff *)malloc(sizeof(zhuff)), *dst = (zhuff *)malloc(sizeof(zhuff));
    size_t o = 0;
    int ok = 0, last = 0;
    zbits z; z.p = data + 2; z.end = data + n; z.buf = 0; z.cnt = 0; z.bad = 0;
    if (!out || !lit || !dst) goto done;
    while (!last) {
        last = (int)zget(&z, 1);
        const int type = (int)zget(&z, 2);
        if (z.bad) goto done;
        if (type == 0) {                                             /* stored */
            z.buf = 0; z.cnt = 0;
            if (z.end - z.p < 4) goto done;
            const unsigned int len = z.p[0] | (z.p[1] << 8), nlen = z.p[2] | (z.p[3] << 8);
            z.p += 4;
            if ((len ^ 0xffffu) != nlen || (size_t)(z.end - z.p) < len) goto done;
            const size_t take = len < want - o ? len : want - o;
            memcpy(out + o, z.p, take); o += take; z.p += len;
            if (o == want) break;
            continue;
        }
        if (type == 3) goto done;
        unsigned char lens[320];
        if (type == 1) {                                             /* fixed codes */
            int i = 0;
            for (; i < 144; i++) lens[i] = 8;
            for (; i < 256; i++) lens[i] = 9;
            for (; i < 280; i++) lens[i] = 7;
            for (; i < 288; i++) lens[i] = 8;
            if (!zhuff_build(lit, lens, 288)) goto done;
            for (i = 0; i < 30; i++) lens[i] = 5;
            if (!zhuff_build(dst, lens, 30)) goto done;
        } else {                                                     /* dynamic codes */
            const int nlen = (int)zget(&z, 5) + 257, ndist = (int)zget(&z, 5) + 1, ncode = (int)zget(&z, 4) + 4;
            if (z.bad || nlen > 286 || ndist > 30) goto done;
            unsigned char cl[19]; memset(cl, 0, sizeof cl);
            for (int i = 0; i < ncode; i++) cl[order[i]] = (unsigned char)zget(&z, 3);
            zhuff clh;
            if (z.bad || !zhuff_build(&clh, cl, 19)) goto done;
            int i = 0;
            while (i < nlen + ndist) {
                const int sym = zhuff_decode(&z, &clh);
                if (sym < 0) goto done;
                if (sym < 16) { lens[i++] = (unsigned char)sym; continue; }
                int rep, val = 0;
                if (sym == 16) { if (i == 0) goto done; val = lens[i - 1]; rep = 3 + (int)zget(&z, 2); }
                else if (sym == 17) rep = 3 + (int)zget(&z, 3);
                else rep = 11 + (int)zget(&z, 7);
                if (z.bad || i + rep > nlen + ndist) goto done;
                while (rep-- > 0) lens[i++] = (unsigned char)val;
            }
            if (lens[256] == 0 || !zhuff_build(lit, lens, nlen) || !zhuff_build(dst, lens + nlen, ndist)) goto done;
        }
        for (;;) {
            int sym = zhuff_decode(&z, lit);
            if (sym < 0) goto done;
            if (sym < 256) { out[o++] = (unsigned char)sym; if (o == want) { last = 1; break; } continue; }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) goto done;
            const size_t len = len_base[sym] + zget(&z, len_extra[sym]);
            const int ds = zhuff_decode(&z, dst);
            if (ds < 0 || ds >= 30) goto done;
            const size_t dist = dist_base[ds] + zget(&z, dist_extra[ds]);
            if (z.bad || dist > o) goto done;
            for (size_t k = 0; k < len && o < want; k++, o++) out[o] = out[o - dist];
            if (o == want) { last = 1; break; }
        }
    }
    ok = o == want;
done:
    free(lit); free(dst);
    if (!ok) { free(out); out = NULL; }
    return out;
}

/* ---- PNG ----------------------------------------------------------------------------------------------------------------------- */
static unsigned int be32(const unsigned char *p) { return ((unsigned int)p[0] << 24) | ((unsigned int)p[1] << 16) | ((unsigned int)p[2] << 8) | p[3]; }
static int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}
/* One (sub-)image of w x h pixels: undo the row filters in place (raw = h rows of 1 + stride bytes), then write its pixels as RGB8 to
 * out[(y0 + y * dy) * W + x0 + x * dx].  Returns 0 on a bad filter byte or palette index. */
static int png_pass(unsigned char *raw, int w, int h, int depth, int color, const unsigned char *pal, int pal_n,
                    unsigned char *out, int W, int x0, int y0, int dx, int dy) {
    const int chan = color == 0 ? 1 : (color == 2 ? 3 : (color == 3 ? 1 : (color == 4 ? 2 : 4)));
    const int bits = chan * depth, bpp = bits >= 8 ? bits / 8 : 1;
    const size_t stride = ((size_t)w * bits + 7) / 8;
    for (int y = 0; y < h; y++) {
        unsigned char *cur = raw + (size_t)y * (stride + 1) + 1;
        const unsigned char *up = y ? cur - (stride + 1) : NULL;
        const int f = cur[-1];
        if (f > 4) return 0;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
            int v = cur[i];
            if (f == 1) v += a; else if (f == 2) v += b; else if (f == 3) v += (a + b) >> 1; else if (f == 4) v += paeth(a, b, c);
            cur[i] = (unsigned char)v;
        }
        unsigned char *o = out + ((size_t)(y0 + y * dy) * W + x0) * 3;
        for (int x = 0; x < w; x++, o += 3 * (size_t)dx) {
            if (depth == 16) {                                        /* big-endian samples: the high byte is the 8-bit value */
                const unsigned char *s = cur + (size_t)x * chan * 2;
                if (color == 0 || color == 4) o[0] = o[1] = o[2] = s[0]; else { o[0] = s[0]; o[1] = s[2]; o[2] = s[4]; }
            } else if (depth == 8) {
                const unsigned char *s = cur + (size_t)x * chan;
                if (color == 3) { if (s[0] >= pal_n) return 0; o[0] = pal[3 * s[0]]; o[1] = pal[3 * s[0] + 1]; o[2] = pal[3 * s[0] + 2]; }
                else if (color == 0 || color == 4) o[0] = o[1] = o[2] = s[0];
                else { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
            } else {                                                  /* 1, 2 or 4 bits: grey or palette, most significant first */
                const int per = 8 / depth, v = (cur[x / per] >> (8 - depth * (x % per + 1))) & ((1 << depth) - 1);
                if (color == 3) { if (v >= pal_n) return 0; o[0] = pal[3 * v]; o[1] = pal[3 * v + 1]; o[2] = pal[3 * v + 2]; }
                else o[0] = o[1] = o[2] = (unsigned char)(v * (depth == 1 ? 255 : (depth == 2 ? 85 : 17)));
            }
        }
    }
    return 1;
}
int rt_png_decode(const unsigned char *data, size_t n, int *width, int *height, unsigned char **rgb_out) {
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (!data || n < 8 + 25 || memcmp(data, sig, 8) != 0) return 0;
    size_t pos = 8, idat_len = 0;
    unsigned int W = 0, H = 0; int depth = 0, color = 0, interlace = 0, have_hdr = 0, pal_n = 0, ok = 0, seen_end = 0;
    unsigned char pal[768];
    unsigned char *idat = NULL, *raw = NULL, *rgb = NULL;
    while (pos + 12 <= n && !seen_end) {
        const unsigned int len = be32(data + pos), type = be32(data + pos + 4);
        if (len > (1u << 30) || pos + 12 + (size_t)len > n) goto done;
        const unsigned char *c = data + pos + 8;
        if (!have_hdr && type != 0x49484452u) goto done;             /* IHDR comes first */
        if (type == 0x49484452u) {                                   /* IHDR */
            if (have_hdr || len != 13) goto done;
            W = be32(c); H = be32(c + 4); depth = c[8]; color = c[9]; interlace = c[12];
            if (W == 0 || H == 0 || W > (1u << 24) || H > (1u << 24) || (unsigned long long)W * H > (1ull << 28)) goto done;
            if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) goto done;
            if (color > 6 || color == 1 || color == 5 || (color == 3 && depth == 16) || ((color == 2 || color == 4 || color == 6) && depth < 8)) goto done;
            if (c[10] != 0 || c[11] != 0 || interlace > 1) goto done;
            have_hdr = 1;
        } else if (type == 0x504c5445u) {                            /* PLTE */
            if (len > 768 || len % 3 != 0 || idat) goto done;
            memcpy(pal, c, len); pal_n = (int)(len / 3);
        } else if (type == 0x49444154u) {                            /* IDAT: the zlib stream, possibly in pieces */
            if (color == 3 && pal_n == 0) goto done;
            unsigned char *grown = (unsigned char *)realloc(idat, idat_len + len + 1);
            if (!grown) goto done;
            idat = grown; memcpy(idat + idat_len, c, len); idat_len += len;
        } else if (type == 0x49454e44u) seen_end = 1;                /* IEND */
        else if (!(type & 0x20000000u) && type != 0x74524e53u) goto done;   /* an unknown CRITICAL chunk (tRNS only adds alpha, which is dropped) */
        pos += 12 + (size_t)len;
    }
    if (!have_hdr || !idat) goto done;
    {
        const int chan = color == 0 ? 1 : (color == 2 ? 3 : (color == 3 ? 1 : (color == 4 ? 2 : 4)));
        static const int xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
        size_t total = 0;
        for (int p = 0; p < (interlace ? 7 : 1); p++) {
            const size_t w = interlace ? (W - xo[p] + xs[p] - 1) / xs[p] : W, h = interlace ? (H - yo[p] + ys[p] - 1) / ys[p] : H;
            if (w && h) total += h * (1 + (w * chan * depth + 7) / 8);
        }
        raw = zlib_inflate(idat, idat_len, total);
        rgb = (unsigned char *)malloc((size_t)W * H * 3);
        if (!raw || !rgb) goto done;
        unsigned char *r = raw;
        for (int p = 0; p < (interlace ? 7 : 1); p++) {
            const size_t w = interlace ? (W - xo[p] + xs[p] - 1) / xs[p] : W, h = interlace ? (H - yo[p] + ys[p] - 1) / ys[p] : H;
            if (!w || !h) continue;
            if (!png_pass(r, (int)w, (int)h, depth, color, pal, pal_n, rgb, (int)W, interlace ? xo[p] : 0, interlace ? yo[p] : 0, interlace ? xs[p] : 1, interlace ? ys[p] : 1)) goto done;
            r += h * (1 + (w * chan * depth + 7) / 8);
        }
    }
    *width = (int)W; *height = (int)H; *rgb_out = rgb; rgb = NULL; ok = 1;
done:
    free(idat); free(raw); free(rgb);
    return ok;
}

/* ---- BMP ----------------------------------------------------------------------------------------------------------------------- */
static unsigned int le16(const unsigned char *p) { return p[0] | ((unsigned int)p[1] << 8); }
static unsigned int le32(const unsigned char *p) { return p[0] | ((unsigned int)p[1] << 8) | ((unsigned int)p[2] << 16) | ((unsigned int)p[3] << 24); }
/* the field of v under a contiguous mask as 8 bits: its n bits repeated until eight are filled */
static int mask_bits(unsigned int m) { int c = 0; for (; m; m >>= 1) c += (int)(m & 1u); return c; }
static unsigned char mask_field(unsigned int v, unsigned int mask) {
    int low = 0; while (!((mask >> low) & 1u)) low++;
    const int n = mask_bits(mask);
    const unsigned int f = (v & mask) >> low;
    unsigned int r = 0;
    for (int s = 8 - n; s > -n; s -= n) r |= s >= 0 ? f << s : f >> -s;
    return (unsigned char)r;
}
int rt_bmp_decode(const unsigned char *data, size_t n, int *width, int *height, unsigned char **rgb_out) {
    if (!data || n < 26 || data[0] != 'B' || data[1] != 'M') return 0;
    const unsigned int offset = le32(data + 10), hsz = le32(data + 14);
    if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) return 0;
    if (14 + (size_t)hsz > n || offset > n) return 0;
    int W, H, bpp; unsigned int compress = 0, mr = 0, mg = 0, mb = 0;
    size_t after = 14 + hsz;                                          /* where the palette starts */
    if (hsz == 12) { W = (int)le16(data + 18); H = (int)le16(data + 20); if (le16(data + 22) != 1) return 0; bpp = (int)le16(data + 24); }
    else {
        W = (int)le32(data + 18); H = (int)le32(data + 22);
        if (le16(data + 26) != 1) return 0;
        bpp = (int)le16(data + 28); compress = le32(data + 30);
        if (compress == 1 || compress == 2) return 0;                 /* RLE */
        if (hsz == 40 || hsz == 56) {
            if (bpp == 16 || bpp == 32) {
                if (compress == 0) { if (bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; } else { mr = 31u << 10; mg = 31u << 5; mb = 31u; } }
                else if (compress == 3) {                             /* three masks BEHIND the header -- also behind a 56-byte one, whose own mask fields stb skips */
                    if (after + 12 > n) return 0;
                    mr = le32(data + after); mg = le32(data + after + 4); mb = le32(data + after + 8);
                    after += 12;
                    if (mr == mg && mg == mb) return 0;
                } else return 0;
            }
        } else { mr = le32(data + 54); mg = le32(data + 58); mb = le32(data + 62); }
    }
    const int flip = H > 0;
    if (H < 0) H = -H;
    if (W <= 0 || H <= 0 || W > (1 << 24) || H > (1 << 24) || (long long)W * H > (1ll << 28)) return 0;
    unsigned char *rgb = (unsigned char *)malloc((size_t)W * H * 3);
    if (!rgb) return 0;
    int ok = 0;
    if (bpp == 1 || bpp == 4 || bpp == 8) {
        const int entry = hsz == 12 ? 3 : 4;
        if ((size_t)offset < after) goto done;
        /* entries of the palette, from the gap between header and pixels; for a 12-byte header stb takes (offset - 14 - 24) / 3, four short */
        const int psize = hsz == 12 ? ((int)offset - 38) / 3 : (int)((offset - after) >> 2);
        if (psize <= 0 || psize > 256) goto done;
        const unsigned char *pal = data + after;
        const size_t rowb = (size_t)(bpp == 1 ? (W + 7) >> 3 : (bpp == 4 ? (W + 1) >> 1 : W)), row = (rowb + 3) & ~(size_t)3;
        if ((size_t)offset + row * (size_t)(H - 1) + rowb > n) goto done;
        for (int y = 0; y < H; y++) {
            const unsigned char *s = data + offset + row * (size_t)y;
            unsigned char *o = rgb + (size_t)(flip ? H - 1 - y : y) * W * 3;
            for (int x = 0; x < W; x++, o += 3) {
                const int v = bpp == 8 ? s[x] : (bpp == 4 ? (s[x >> 1] >> ((x & 1) ? 0 : 4)) & 15 : (s[x >> 3] >> (7 - (x & 7))) & 1);
                if (v >= psize) goto done;                            /* (stb reads its uninitialised palette here) */
                o[0] = pal[v * entry + 2]; o[1] = pal[v * entry + 1]; o[2] = pal[v * entry];
            }
        }
        ok = 1;
    } else if (bpp == 24 || bpp == 16 || bpp == 32) {
        if (bpp != 24 && (!mr || !mg || !mb || mask_bits(mr) > 8 || mask_bits(mg) > 8 || mask_bits(mb) > 8)) goto done;
        const size_t rowb = (size_t)W * (size_t)(bpp / 8), row = (rowb + 3) & ~(size_t)3;
        if ((size_t)offset != after || (size_t)offset + row * (size_t)(H - 1) + rowb > n) goto done;   /* no palette: the pixels follow the header at once */
        for (int y = 0; y < H; y++) {
            const unsigned char *s = data + offset + row * (size_t)y;
            unsigned char *o = rgb + (size_t)(flip ? H - 1 - y : y) * W * 3;
            for (int x = 0; x < W; x++, o += 3) {
                if (bpp == 24) { o[0] = s[3 * x + 2]; o[1] = s[3 * x + 1]; o[2] = s[3 * x]; continue; }
                const unsigned int v = bpp == 16 ? le16(s + 2 * x) : le32(s + 4 * x);
                o[0] = mask_field(v, mr); o[1] = mask_field(v, mg); o[2] = mask_field(v, mb);
            }
        }
        ok = 1;
    }
done:
    if (ok) { *width = W; *height = H; *rgb_out = rgb; } else free(rgb);
    return ok;
}
