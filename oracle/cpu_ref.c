/*
 * oracle/cpu_ref.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Double-precision restatement of the reference's path-tracing hot path
 * (see cpu_ref.h for scope, pinning and the rules about who may load it).
 * Every routine cites the reference file:line (under /root/reference/)
 * whose arithmetic ORDER it reproduces: the goal is bit-identical doubles,
 * so expressions are written in the reference's association order and the
 * file must be built with -ffp-contract=off.
 */
#include "cpu_ref.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* stage + counters                                                    */
/* ------------------------------------------------------------------ */
static int g_stage = 3;
void orc_set_stage(int stage) { g_stage = stage; }
int  orc_get_stage(void) { return g_stage; }

static _Thread_local orc_counters t_cnt;
static orc_counters g_cnt;
#define CNT(f) (t_cnt.f++)
static void counters_flush(void) {
    uint64_t *g = (uint64_t *)&g_cnt, *t = (uint64_t *)&t_cnt;
#pragma omp critical(orc_cnt)
    for (size_t i = 0; i < sizeof(orc_counters) / sizeof(uint64_t); ++i) { g[i] += t[i]; t[i] = 0; }
}
void orc_counters_reset(void) { memset(&g_cnt, 0, sizeof g_cnt); memset(&t_cnt, 0, sizeof t_cnt); }
void orc_counters_get(orc_counters *out) { counters_flush(); *out = g_cnt; }

/* ------------------------------------------------------------------ */
/* RNG: utils.h:23-26                                                  */
/* ------------------------------------------------------------------ */
typedef struct { int mode; uint32_t key[2]; uint32_t ctr[4]; uint32_t buf[4]; int left; } rng_state;
static _Thread_local rng_state t_rng;

static inline void mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo) {
    uint64_t p = (uint64_t)a * b; *hi = (uint32_t)(p >> 32); *lo = (uint32_t)p;
}
/* Philox4x32-10, Salmon et al. SC'11 (constants of the Random123 distribution). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void orc_rng_mode(int mode) { t_rng.mode = mode; t_rng.left = 0; }
void orc_rng_seek(uint64_t seed, uint32_t pixel, uint32_t sample) {
    t_rng.key[0] = (uint32_t)seed; t_rng.key[1] = (uint32_t)(seed >> 32);
    t_rng.ctr[0] = pixel; t_rng.ctr[1] = sample; t_rng.ctr[2] = 0; t_rng.ctr[3] = 0x0C0FFEE0u; /* oracle stream tag */
    t_rng.left = 0;
}
double random_double(void) {
    if (t_rng.mode == ORC_RNG_GLIBC) return rand() / (RAND_MAX + 1.0);
    if (t_rng.left == 0) { orc_philox4x32_10(t_rng.ctr, t_rng.key, t_rng.buf); t_rng.ctr[2]++; t_rng.left = 4; }
    return t_rng.buf[4 - t_rng.left--] * (1.0 / 4294967296.0);
}
double random_double_scaled(double min, double max) { return min + (max - min) * random_double(); }
int random_int(int min, int max) { return (int)random_double_scaled(min, max + 1); }
double clamp(double x, double min, double max) { return (x < min) ? min : ((x > max) ? max : x); }
double deg2rad(double deg) { return deg * PI / 180.0; }

/* ------------------------------------------------------------------ */
/* vec3: vec3.c:15-107 (left-to-right sums, true division in div_k)    */
/* ------------------------------------------------------------------ */
static inline vec3 V(double x, double y, double z) { vec3 r = {x, y, z}; return r; }
static inline vec3 vadd(vec3 a, vec3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 vsub(vec3 a, vec3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 vmul(vec3 a, vec3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 vscale(vec3 a, double k) { return V(a.x * k, a.y * k, a.z * k); }
/* vec3.c:19,33: "v->x * 1/k" parses as (v->x*1)/k, an IEEE division (F12) */
static inline vec3 vdiv(vec3 a, double k) { return V(a.x * 1 / k, a.y * 1 / k, a.z * 1 / k); }
static inline double vdot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline double vlen2(vec3 a) { return (a.x * a.x) + (a.y * a.y) + (a.z * a.z); }
static inline double vlen(vec3 a) { return sqrt(vlen2(a)); }
static inline vec3 vcross(vec3 a, vec3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline vec3 vunit(vec3 a) { return vdiv(a, vlen(a)); }
static inline vec3 vneg(vec3 a) { return V(-a.x, -a.y, -a.z); }

vec3 vec3c_reflect(vec3 v, vec3 n) { double a = vdot(v, n) * 2.0; return vsub(v, vscale(n, a)); } /* vec3.c:26,40 */
vec3 vec3c_refract(vec3 uv, vec3 n, double etai_over_etat) { /* vec3.c:44-50 */
    double cos_theta = fmin(vdot(vneg(uv), n), 1.0);
    vec3 perp = vscale(vadd(uv, vscale(n, cos_theta)), etai_over_etat);
    double k = -sqrt(fabs(1.0 - vlen2(perp)));
    return vadd(perp, vscale(n, k));
}
/* gcc evaluates the three initialisers left to right (SURVEY 8c.4 caveat) */
vec3 vec3_random(void) { double a = random_double(), b = random_double(), c = random_double(); return V(a, b, c); } /* vec3.c:56-58 */
vec3 vec3_random_scaled(double min, double max) { /* vec3.c:61-63 */
    double a = random_double_scaled(min, max), b = random_double_scaled(min, max), c = random_double_scaled(min, max);
    return V(a, b, c);
}
vec3 vec3_random_in_unit_sphere(void) { /* vec3.c:66-72 */
    for (;;) { vec3 p = vec3_random_scaled(-1, 1); if (vlen2(p) >= 1) continue; return p; }
}
vec3 vec3_random_unit_vector_in_unit_sphere(void) { return vunit(vec3_random_in_unit_sphere()); } /* vec3.c:75-77 */
vec3 random_in_unit_disk(void) { /* vec3.c:87-93 */
    for (;;) {
        double a = random_double_scaled(-1, 1), b = random_double_scaled(-1, 1);
        vec3 p = V(a, b, 0);
        if (vlen2(p) >= 1) continue;
        return p;
    }
}
vec3 random_cosine_direction(void) { /* vec3.c:97-107 */
    double r1 = random_double(), r2 = random_double();
    double z = sqrt(1 - r2);
    double phi = 2 * PI * r1;
    double x = cos(phi) * sqrt(r2), y = sin(phi) * sqrt(r2);
    return V(x, y, z);
}
static int near_zero(vec3 v) { return (fabs(v.x) < 1e-8) && (fabs(v.y) < 1e-8) && (fabs(v.z) < 1e-8); } /* vec3.c:53 */

point3 ray_at(ray *r, double t) { return vadd(r->orig, vscale(r->dir, t)); } /* ray.c:8-11 */

/* ------------------------------------------------------------------ */
/* registries: growable (the reference's are fixed 1024-slot arrays, F17) */
/* ------------------------------------------------------------------ */
typedef struct { void **v; int n, cap; } ptrvec;
static void pv_push(ptrvec *p, void *x) {
    if (p->n == p->cap) { p->cap = p->cap ? p->cap * 2 : 64; p->v = realloc(p->v, sizeof(void *) * (size_t)p->cap); }
    p->v[p->n++] = x;
}
static ptrvec g_lists, g_mats, g_texs;

/* ------------------------------------------------------------------ */
/* perlin: perlin.c:13-148                                             */
/* ------------------------------------------------------------------ */
static int *perlin_perm(void) { /* perlin.c:13-29 */
    int *p = malloc(sizeof(int) * POINT_COUNT);
    for (int i = 0; i < POINT_COUNT; ++i) p[i] = i;
    for (int i = 0; i < POINT_COUNT; ++i) { int target = random_int(0, i); int tmp = p[i]; p[i] = p[target]; p[target] = tmp; }
    return p;
}
perlin *perlin_init(void) { /* perlin.c:67-75 */
    perlin *p = malloc(sizeof(perlin));
    p->ranvec = malloc(sizeof(vec3) * POINT_COUNT);
    for (int i = 0; i < POINT_COUNT; ++i) p->ranvec[i] = vunit(vec3_random_scaled(-1, 1));
    p->perm_x = perlin_perm(); p->perm_y = perlin_perm(); p->perm_z = perlin_perm();
    return p;
}
double perlin_noise(perlin *p, point3 pt) { /* perlin.c:108-128 + interp 33-50 */
    double u = pt.x - floor(pt.x), v = pt.y - floor(pt.y), w = pt.z - floor(pt.z);
    int i = (int)floor(pt.x), j = (int)floor(pt.y), k = (int)floor(pt.z);
    double uu = u * u * (3 - 2 * u), vv = v * v * (3 - 2 * v), ww = w * w * (3 - 2 * w);
    double accum = 0.0;
    for (int di = 0; di < 2; di++) for (int dj = 0; dj < 2; dj++) for (int dk = 0; dk < 2; dk++) {
        vec3 c = p->ranvec[p->perm_x[(i + di) & 255] ^ p->perm_y[(j + dj) & 255] ^ p->perm_z[(k + dk) & 255]];
        vec3 wv = V(u - di, v - dj, w - dk);
        accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) * (dk * ww + (1 - dk) * (1 - ww)) * vdot(c, wv);
    }
    return accum;
}
double perlin_turb(perlin *perl, point3 p) { /* perlin.c:132-148, depth 7 */
    double accum = 0.0, weight = 1.0; point3 tp = p;
    for (int i = 0; i < 7; i++) { accum += weight * perlin_noise(perl, tp); weight *= 0.5; tp = vscale(tp, 2); }
    return fabs(accum);
}

/* ------------------------------------------------------------------ */
/* textures: texture.c:14-166                                          */
/* ------------------------------------------------------------------ */
static texture *tex_new(texture_type type, void *o) { texture *t = malloc(sizeof *t); t->type = type; t->o = o; pv_push(&g_texs, t); return t; }
texture *texture_solid_color_init(color c) { solid_color *s = malloc(sizeof *s); s->color_value = c; return tex_new(TEXTURE_SOLID, s); }
texture *texture_solid_color_init_rgb(double r, double g, double b) { return texture_solid_color_init(V(r, g, b)); }
texture *texture_checker_init(texture *odd, texture *even) { checker *c = malloc(sizeof *c); c->odd = odd; c->even = even; return tex_new(TEXTURE_CHECKER, c); }
texture *texture_checker_init_c(color odd, color even) { texture *o = texture_solid_color_init(odd); texture *e = texture_solid_color_init(even); return texture_checker_init(o, e); }
texture *texture_noise_init_scaled(double sc) { texture_noise *n = malloc(sizeof *n); n->noise = perlin_init(); n->scale = sc; return tex_new(TEXTURE_NOISE, n); }
texture *texture_noise_init(void) { return texture_noise_init_scaled(1.0); }
texture *orc_texture_image_from_rgb(const unsigned char *rgb, int w, int h) {
    texture_image *im = malloc(sizeof *im);
    im->width = w; im->height = h; im->bytes_per_scanline = 3 * w;
    im->data = malloc((size_t)w * h * 3); memcpy(im->data, rgb, (size_t)w * h * 3);
    return tex_new(TEXTURE_IMAGE, im);
}
texture *texture_image_init(char *filename) { /* texture.c:63-70; the oracle reads binary PPM instead of stb_image formats */
    FILE *f = fopen(filename, "rb"); int w, h, mx;
    if (!f || fscanf(f, "P6 %d %d %d", &w, &h, &mx) != 3 || mx != 255) { printf("!! could not load texture image file '%s'! \n", filename); if (f) fclose(f); return NULL; }
    fgetc(f);
    unsigned char *buf = malloc((size_t)w * h * 3);
    size_t got = fread(buf, 1, (size_t)w * h * 3, f); fclose(f);
    texture *t = got == (size_t)w * h * 3 ? orc_texture_image_from_rgb(buf, w, h) : NULL;
    free(buf); return t;
}
void textures_free_all(void) {
    for (int i = 0; i < g_texs.n; i++) {
        texture *t = g_texs.v[i];
        if (t->type == TEXTURE_NOISE) { perlin *p = ((texture_noise *)t->o)->noise; free(p->ranvec); free(p->perm_x); free(p->perm_y); free(p->perm_z); free(p); }
        else if (t->type == TEXTURE_IMAGE) free(((texture_image *)t->o)->data);
        free(t->o); free(t);
    }
    g_texs.n = 0;
}
color texture_value(texture *t, double u, double v, vec3 *p) { /* texture.c:115-166 */
    switch (t->type) {
    case TEXTURE_SOLID: return ((solid_color *)t->o)->color_value;
    case TEXTURE_CHECKER: {
        CNT(tex_checker);
        checker *c = t->o;
        double sines = sin(10 * p->x) * sin(10 * p->y) * sin(10 * p->z);
        return (sines < 0) ? texture_value(c->odd, u, v, p) : texture_value(c->even, u, v, p);
    }
    case TEXTURE_NOISE: {
        CNT(tex_noise);
        texture_noise *tn = t->o;
        double perl = perlin_turb(tn->noise, *p);
        double val = 0.5 * (1.0 + sin(tn->scale * p->z + 10 * perl));
        return vscale(V(1, 1, 1), val);
    }
    case TEXTURE_IMAGE: {
        CNT(tex_image);
        texture_image *img = t->o;
        if (!img->data) return V(1, 0, 1);
        u = clamp(u, 0.0, 1.0); v = 1.0 - clamp(v, 0.0, 1.0);
        int i = (int)(u * img->width), j = (int)(v * img->height);
        if (i >= img->width) i = img->width - 1;
        if (j >= img->height) j = img->height - 1;
        const double color_scale = 1.0 / 255.0;
        unsigned char *px = img->data + j * img->bytes_per_scanline + i * 3;
        return V(color_scale * px[0], color_scale * px[1], color_scale * px[2]);
    }
    }
    printf("texture_value not implemented for type %d\n", t->type); return V(0, 0, 0);
}

/* ------------------------------------------------------------------ */
/* onb: onb.c:7-30                                                     */
/* ------------------------------------------------------------------ */
void onb_init_from_w(onb *o, vec3 *n) {
    CNT(onb_builds);
    o->axis[2] = vunit(*n);
    vec3 a = (fabs(o->axis[2].x) > 0.9) ? V(0, 1, 0) : V(1, 0, 0);
    o->axis[1] = vunit(vcross(o->axis[2], a));
    o->axis[0] = vcross(o->axis[2], o->axis[1]);
}
vec3 onb_localv(onb *o, vec3 a) { /* u*a.x + (v*a.y + w*a.z) */
    return vadd(vscale(o->axis[0], a.x), vadd(vscale(o->axis[1], a.y), vscale(o->axis[2], a.z)));
}

/* ------------------------------------------------------------------ */
/* aabb: aabb.c:33-84 (float reciprocal, F9)                           */
/* ------------------------------------------------------------------ */
int aabb_hit(aabb *b, ray *r, double tmin, double tmax) {
    CNT(aabb_tests);
    const double mn[3] = {b->minimum.x, b->minimum.y, b->minimum.z}, mx[3] = {b->maximum.x, b->maximum.y, b->maximum.z};
    const double o[3] = {r->orig.x, r->orig.y, r->orig.z}, d[3] = {r->dir.x, r->dir.y, r->dir.z};
    for (int a = 0; a < 3; a++) {
        float invD = 1.0f / (float)d[a];
        double t0 = (mn[a] - o[a]) * invD, t1 = (mx[a] - o[a]) * invD;
        if (invD < 0.0f) { double t = t0; t0 = t1; t1 = t; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return 0;
    }
    return 1;
}
aabb surrounding_box(aabb a, aabb b) {
    aabb r = {V(fmin(a.minimum.x, b.minimum.x), fmin(a.minimum.y, b.minimum.y), fmin(a.minimum.z, b.minimum.z)),
              V(fmax(a.maximum.x, b.maximum.x), fmax(a.maximum.y, b.maximum.y), fmax(a.maximum.z, b.maximum.z))};
    return r;
}

/* ------------------------------------------------------------------ */
/* materials: material.c:35-245 (stage 3), 02 src/material.c:120-218   */
/* ------------------------------------------------------------------ */
static material *mat_new(material_type t, void *payload) { material *m = malloc(sizeof *m); m->type = t; m->mat = payload; pv_push(&g_mats, m); return m; }
material *material_lambertian_new(texture *t) { material_lambertian *l = malloc(sizeof *l); l->albedo = t; return mat_new(MATERIAL_LAMBERTIAN, l); }
material *material_lambertian_new_from_color(color c) { return material_lambertian_new(texture_solid_color_init(c)); }
material *material_metal_new(color a, double fuzz) { material_metal *m = malloc(sizeof *m); m->albedo = a; m->fuzz = fuzz; return mat_new(MATERIAL_METAL, m); }
material *material_dielectric_new(double ir) { material_dielectric *d = malloc(sizeof *d); d->ir = ir; return mat_new(MATERIAL_DIELECTRIC, d); }
material *material_light_new(texture *t) { material_light *l = malloc(sizeof *l); l->emit = t; return mat_new(MATERIAL_LIGHT, l); }
material *material_light_new_from_color(color c) { return material_light_new(texture_solid_color_init(c)); }
material *material_isotropic_new(texture *t) { material_isotropic *l = malloc(sizeof *l); l->albedo = t; return mat_new(MATERIAL_ISOTROPIC, l); }
material *material_isotropic_new_from_color(color c) { return material_isotropic_new(texture_solid_color_init(c)); }
void materials_free_all(void) { for (int i = 0; i < g_mats.n; i++) { material *m = g_mats.v[i]; free(m->mat); free(m); } g_mats.n = 0; }

double reflectance(double cosine, double ref_idx) { /* material.c:142 */
    double r0 = (1 - ref_idx) / (1 + ref_idx); r0 = r0 * r0; return r0 + (1 - r0) * pow((1 - cosine), 5);
}
/* shared by all stages: material.c:144-157 == 02 material.c:147-160 */
static vec3 dielectric_direction(material_dielectric *mat, ray *r, hit_record *rec) {
    double ratio = rec->front_face ? (1.0 / mat->ir) : mat->ir;
    vec3 ud = vunit(r->dir);
    double cos_theta = fmin(vdot(vneg(ud), rec->normal), 1.0);
    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    int cannot_refract = ratio * sin_theta > 1.0;
    if (cannot_refract || reflectance(cos_theta, ratio) > random_double()) return vec3c_reflect(ud, rec->normal);
    return vec3c_refract(ud, rec->normal, ratio);
}
int material_scatter(material *mat, ray *r, hit_record *rec, scatter_record *srec) { /* material.c:121-192 */
    switch (mat->type) {
    case MATERIAL_LAMBERTIAN:
        srec->attenuation = texture_value(((material_lambertian *)mat->mat)->albedo, rec->u, rec->v, &rec->p);
        srec->is_specular = 0;
        srec->pdf_ptr = pdf_cosine_init(&rec->normal);
        return 1;
    case MATERIAL_METAL: { /* draws the unit-sphere sample even when fuzz==0 (F13); always returns 1 (F14) */
        material_metal *m = mat->mat;
        vec3 reflected = vec3c_reflect(vunit(r->dir), rec->normal);
        ray sr = {rec->p, vadd(reflected, vscale(vec3_random_in_unit_sphere(), m->fuzz)), r->time};
        srec->specular_ray = sr; srec->attenuation = m->albedo; srec->is_specular = 1; srec->pdf_ptr = NULL;
        return 1;
    }
    case MATERIAL_DIELECTRIC: {
        ray sr = {rec->p, dielectric_direction(mat->mat, r, rec), r->time};
        srec->specular_ray = sr; srec->attenuation = V(1.0, 1.0, 1.0); srec->is_specular = 1; srec->pdf_ptr = NULL;
        return 1;
    }
    case MATERIAL_LIGHT: return 0;
    case MATERIAL_ISOTROPIC: return 1; /* material.c:172-176: returns 1 without filling srec (F14); callers must not rely on it */
    }
    printf("material_scatter nor implemented for material type %d\n", mat->type); return 0;
}
int material_scatter_att(material *mat, ray *r, hit_record *rec, color *attenuation, ray *scattered) { /* 02 material.c:120-193; 01 material.h:66-123 */
    switch (mat->type) {
    case MATERIAL_LAMBERTIAN: {
        vec3 dir = vadd(rec->normal, vec3_random_unit_vector_in_unit_sphere());
        if (near_zero(dir)) dir = rec->normal;
        ray s = {rec->p, dir, r->time}; *scattered = s;
        *attenuation = texture_value(((material_lambertian *)mat->mat)->albedo, rec->u, rec->v, &rec->p);
        return 1;
    }
    case MATERIAL_METAL: {
        material_metal *m = mat->mat;
        vec3 reflected = vec3c_reflect(vunit(r->dir), rec->normal);
        vec3 dir = vadd(reflected, vscale(vec3_random_in_unit_sphere(), m->fuzz));
        ray s = {rec->p, dir, r->time}; *scattered = s; *attenuation = m->albedo;
        return vdot(scattered->dir, rec->normal) > 0;
    }
    case MATERIAL_DIELECTRIC: {
        ray s = {rec->p, dielectric_direction(mat->mat, r, rec), r->time}; *scattered = s; *attenuation = V(1, 1, 1);
        return 1;
    }
    case MATERIAL_LIGHT: return 0;
    case MATERIAL_ISOTROPIC: {
        ray s = {rec->p, vec3_random_in_unit_sphere(), r->time}; *scattered = s;
        *attenuation = texture_value(((material_isotropic *)mat->mat)->albedo, rec->u, rec->v, &rec->p);
        return 1;
    }
    }
    printf("material_scatter nor implemented for material type %d\n", mat->type); return 0;
}
double material_scattering_pdf(material *mat, ray *r, hit_record *rec, ray *scattered) { /* material.c:205-216 */
    (void)r;
    if (mat->type == MATERIAL_LAMBERTIAN) { double cosine = vdot(rec->normal, vunit(scattered->dir)); return cosine < 0 ? 0 : cosine / PI; }
    printf("material_scattering_pdf nor implemented for material type %d\n", mat->type); return 0;
}
color material_emitted(material *mat, hit_record *rec, double u, double v, point3 p) { /* material.c:229-245; 02: two-sided (F15) */
    if (mat->type != MATERIAL_LIGHT) return V(0, 0, 0);
    if (g_stage == 3 && !rec->front_face) return V(0, 0, 0);
    return texture_value(((material_light *)mat->mat)->emit, u, v, &p);
}

/* ------------------------------------------------------------------ */
/* hittable_list: hittable_list.c:11-180                                */
/* ------------------------------------------------------------------ */
hittable_list *hittable_list_new(int max_size) {
    hittable_list *l = malloc(sizeof *l);
    if (max_size < 4) max_size = 4;
    l->objs = malloc(sizeof(hittable *) * (size_t)max_size); l->index = 0; l->size = max_size; l->gc = 1;
    pv_push(&g_lists, l); return l;
}
hittable_list *hittable_list_new_no_gc(int max_size) { hittable_list *l = hittable_list_new(max_size); l->gc = 0; return l; }
void hittable_list_add(hittable_list *l, hittable *o) { /* reference has no bounds check (F17); the oracle grows */
    if (l == NULL) return;
    if (l->index == l->size) { l->size *= 2; l->objs = realloc(l->objs, sizeof(hittable *) * (size_t)l->size); }
    l->objs[l->index++] = o;
}
int hittable_list_index(hittable_list *l) { return l->index; }
void *hittable_list_objects(hittable_list *l) { return (void *)l->objs; }
void hittable_lists_free_all(void) {
    for (int i = 0; i < g_lists.n; i++) {
        hittable_list *l = g_lists.v[i];
        if (l->gc) for (int k = 0; k < l->index; k++) hittable_free(l->objs[k]);
        free(l->objs); free(l);
    }
    g_lists.n = 0;
}
int hittable_list_hit(hittable_list *l, ray *r, double tmin, double tmax, hit_record *rec) { /* :125-143 ties: later object wins */
    hit_record tmp; int hit_anything = 0; double closest = tmax;
    for (int i = 0; i < l->index; ++i)
        if (hittable_hit(l->objs[i], r, tmin, closest, &tmp)) { hit_anything = 1; closest = tmp.t; *rec = tmp; }
    return hit_anything;
}
int hittable_list_bounding_box(hittable_list *l, double t0, double t1, aabb *out) { /* :147-161 */
    if (l->index == 0) return 0;
    aabb tmp; int first = 1;
    for (int i = 0; i < l->index; ++i) {
        if (hittable_bounding_box(l->objs[i], t0, t1, &tmp) == 0) return 0;
        *out = first ? tmp : surrounding_box(*out, tmp); first = 0;
    }
    return 1;
}
double hittable_list_pdf_value(hittable_list *l, point3 *orig, vec3 *v) { /* :165-173 */
    double weight = 1.0 / (double)l->index, sum = 0.0;
    for (int i = 0; i < l->index; ++i) sum += weight * hittable_pdf_value(l->objs[i], *orig, *v);
    return sum;
}
vec3 hittable_list_random(hittable_list *l, vec3 orig) { int n = l->index; return hittable_random(l->objs[random_int(0, n - 1)], orig); } /* :177-180 */

/* ------------------------------------------------------------------ */
/* hittable constructors: hittable.c:87-312                             */
/* ------------------------------------------------------------------ */
static hittable *wrap(hittable_list *world, hittable_type t, void *obj) { hittable *h = malloc(sizeof *h); h->t = t; h->obj = obj; hittable_list_add(world, h); return h; }
hittable *hittable_sphere_new(hittable_list *world, point3 center, double r, material *mat) {
    sphere *s = malloc(sizeof *s); s->center = center; s->r = r; s->mat = mat; return wrap(world, HITTABLE_SPHERE, s);
}
hittable *hittable_moving_sphere_new(hittable_list *world, point3 c0, point3 c1, double t0, double t1, double r, material *mat) {
    moving_sphere *s = malloc(sizeof *s); s->center0 = c0; s->center1 = c1; s->time0 = t0; s->time1 = t1; s->r = r; s->mat = mat;
    return wrap(world, HITTABLE_MOVING_SPHERE, s);
}
hittable *hittable_rect_new(hittable_list *world, double x0, double x1, double y0, double y1, double z0, double z1, double k, rect_axis axis, material *mat) {
    rect *r = malloc(sizeof *r); r->x0 = x0; r->x1 = x1; r->y0 = y0; r->y1 = y1; r->z0 = z0; r->z1 = z1; r->k = k; r->axis = axis; r->mat = mat;
    return wrap(world, HITTABLE_RECT, r);
}
hittable *hittable_box_new(hittable_list *world, point3 mn, point3 mx, material *mat) { /* side order :146-151 */
    box *b = malloc(sizeof *b); b->box_min = mn; b->box_max = mx; b->sides = hittable_list_new(6);
    hittable_rect_new(b->sides, mn.x, mx.x, mn.y, mx.y, 0, 0, mx.z, XY, mat);
    hittable_rect_new(b->sides, mn.x, mx.x, mn.y, mx.y, 0, 0, mn.z, XY, mat);
    hittable_rect_new(b->sides, mn.x, mx.x, 0, 0, mn.z, mx.z, mx.y, XZ, mat);
    hittable_rect_new(b->sides, mn.x, mx.x, 0, 0, mn.z, mx.z, mn.y, XZ, mat);
    hittable_rect_new(b->sides, 0, 0, mn.y, mx.y, mn.z, mx.z, mx.x, YZ, mat);
    hittable_rect_new(b->sides, 0, 0, mn.y, mx.y, mn.z, mx.z, mn.x, YZ, mat);
    return wrap(world, HITTABLE_BOX, b);
}
/* BVH (:163-225, F8): split [start,end) at the midpoint of the ORIGINAL order; the qsort of the private copy
 * is discarded, so the tree shape depends only on n and on the min.x order of 2-element leaves; one rand()
 * per node ("int raxis = random_double()" is always 0 => x comparator). */
static hittable *bvh_build(hittable **src, int start, int end, double t0, double t1) {
    bvh_node *node = malloc(sizeof *node);
    (void)random_double();
    int span = end - start;
    if (span == 1) { node->left = node->right = src[start]; }
    else if (span == 2) {
        aabb a, b;
        int lt = 0;
        if (!hittable_bounding_box(src[start], 0, 0, &a) || !hittable_bounding_box(src[start + 1], 0, 0, &b)) { printf("Failed to get bounding box in bvh_node comparator.\n"); lt = 0; }
        else lt = a.minimum.x < b.minimum.x;
        node->left = src[lt ? start : start + 1]; node->right = src[lt ? start + 1 : start];
    } else {
        int mid = start + span / 2;
        node->left = bvh_build(src, start, mid, t0, t1);
        node->right = bvh_build(src, mid, end, t0, t1);
    }
    aabb bl, br;
    if (!hittable_bounding_box(node->left, t0, t1, &bl) || !hittable_bounding_box(node->right, t0, t1, &br)) { printf("Failed to get bounding box in bvh_node constructor.\n"); return NULL; }
    node->box = surrounding_box(bl, br);
    hittable *h = malloc(sizeof *h); h->t = HITTABLE_BVH_NODE; h->obj = node; return h;
}
hittable *bvh_node_init(hittable_list *world, hittable_list *objects, double t0, double t1) {
    hittable *o = bvh_build(objects->objs, 0, objects->index, t0, t1); hittable_list_add(world, o); return o;
}
hittable *hittable_translate_init(hittable_list *world, hittable *obj, vec3 offset) { translate *t = malloc(sizeof *t); t->obj = obj; t->offset = offset; return wrap(world, HITTABLE_TRANSLATE, t); }
void rotatePoint(vec3 o, double s, double c, rotation_axis axis, vec3 *out) { /* :37-48; axis naming F10; only two components are written */
    if (axis == X) { out->x = c * o.x - s * o.y; out->y = s * o.x + c * o.y; }
    else if (axis == Y) { out->x = c * o.x + s * o.z; out->z = -s * o.x + c * o.z; }
    else { out->y = c * o.y - s * o.z; out->z = s * o.y + c * o.z; }
}
hittable *hittable_rotate_init(hittable_list *world, hittable *obj, double angle, rotation_axis axis) { /* :246-288 */
    rotate *ro = malloc(sizeof *ro); ro->obj = obj; ro->axis = axis;
    double radians = deg2rad(angle);
    ro->sin_theta = sin(radians); ro->cos_theta = cos(radians);
    ro->hasbox = hittable_bounding_box(obj, 0, 1, &ro->bbox);
    point3 mn = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, mx = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++) {
        double x = i * ro->bbox.maximum.x + (1 - i) * ro->bbox.minimum.x;
        double y = j * ro->bbox.maximum.y + (1 - j) * ro->bbox.minimum.y;
        double z = k * ro->bbox.maximum.z + (1 - k) * ro->bbox.minimum.z;
        vec3 p = {x, y, z}, q = p;
        rotatePoint(p, ro->sin_theta, ro->cos_theta, ro->axis, &q);
        mn.x = fmin(mn.x, q.x); mx.x = fmax(mx.x, q.x);
        mn.y = fmin(mn.y, q.y); mx.y = fmax(mx.y, q.y);
        mn.z = fmin(mn.z, q.z); mx.z = fmax(mx.z, q.z);
    }
    ro->bbox.minimum = mn; ro->bbox.maximum = mx;
    return wrap(world, HITTABLE_ROTATE, ro);
}
hittable *hittable_constant_medium_init(hittable_list *world, hittable *b, double d, texture *a) {
    constant_medium *m = malloc(sizeof *m); m->boundary = b; m->neg_inv_density = -1 / d; m->phase_function = material_isotropic_new(a);
    return wrap(world, HITTABLE_CONSTANT_MEDIUM, m);
}
hittable *hittable_constant_medium_init_c(hittable_list *world, hittable *b, double d, color c) { return hittable_constant_medium_init(world, b, d, texture_solid_color_init(c)); }
hittable *hittable_flip_face_init(hittable_list *world, hittable *obj) { flip_face *f = malloc(sizeof *f); f->obj = obj; return wrap(world, HITTABLE_FLIP_FACE, f); }
void hittable_free(hittable *o) { /* :336-366 */
    switch (o->t) {
    case HITTABLE_BVH_NODE: { bvh_node *n = o->obj; if (n->left == n->right) hittable_free(n->left); else { hittable_free(n->left); hittable_free(n->right); } break; }
    case HITTABLE_TRANSLATE: hittable_free(((translate *)o->obj)->obj); break;
    case HITTABLE_ROTATE: hittable_free(((rotate *)o->obj)->obj); break;
    case HITTABLE_CONSTANT_MEDIUM: hittable_free(((constant_medium *)o->obj)->boundary); break;
    case HITTABLE_FLIP_FACE: hittable_free(((flip_face *)o->obj)->obj); break;
    default: break;
    }
    free(o->obj); free(o);
}

/* ------------------------------------------------------------------ */
/* hit tests: hittable.c:13-16, 381-615                                 */
/* ------------------------------------------------------------------ */
static inline void set_facenormal(hit_record *rec, ray *r, vec3 n) { rec->front_face = vdot(r->dir, n) < 0; rec->normal = rec->front_face ? n : vneg(n); }
static void sphere_uv(vec3 p, double *u, double *v) { double theta = acos(-p.y); double phi = atan2(-p.z, p.x) + PI; *u = phi / (2 * PI); *v = theta / PI; }
static int line_sphere(vec3 oc, vec3 dir, double rad, double tmin, double tmax, double *root) { /* :406-429 */
    double a = vlen2(dir), half_b = vdot(oc, dir), c = vlen2(oc) - rad * rad;
    double disc = half_b * half_b - a * c;
    if (disc < 0) return 0;
    double sq = sqrt(disc);
    double t = (-half_b - sq) / a;
    if (t < tmin || t > tmax) { t = (-half_b + sq) / a; if (t < tmin || t > tmax) return 0; }
    *root = t; return 1;
}
static int sphere_hit_center(point3 center, double rad, material *mat, ray *r, double tmin, double tmax, hit_record *rec) {
    double root;
    if (!line_sphere(vsub(r->orig, center), r->dir, rad, tmin, tmax, &root)) return 0;
    rec->t = root; rec->p = ray_at(r, rec->t);
    vec3 n = vdiv(vsub(rec->p, center), rad);
    set_facenormal(rec, r, n); rec->mat = mat; sphere_uv(n, &rec->u, &rec->v);
    return 1;
}
int sphere_hit(sphere *s, ray *r, double tmin, double tmax, hit_record *rec) { /* :433-451 */
    CNT(sphere_tests);
    int h = sphere_hit_center(s->center, s->r, s->mat, r, tmin, tmax, rec); if (h) CNT(sphere_hits); return h;
}
static point3 msphere_center(moving_sphere *s, double t) { /* :455-458 */
    double k = ((t - s->time0) / (s->time1 - s->time0));
    return vadd(vscale(vsub(s->center1, s->center0), k), s->center0);
}
int rect_hit(rect *rc, ray *r, double tmin, double tmax, hit_record *rec) { /* :494-517; NaN t passes the range test (F20) */
    CNT(rect_tests);
    double t;
    if (rc->axis == XY) t = (rc->k - r->orig.z) / r->dir.z;
    else if (rc->axis == XZ) t = (rc->k - r->orig.y) / r->dir.y;
    else t = (rc->k - r->orig.x) / r->dir.x;
    if (t < tmin || t > tmax) return 0;
    double x = r->orig.x + t * r->dir.x, y = r->orig.y + t * r->dir.y, z = r->orig.z + t * r->dir.z;
    if ((rc->axis == XZ && (x < rc->x0 || x > rc->x1 || z < rc->z0 || z > rc->z1)) ||
        (rc->axis == XY && (x < rc->x0 || x > rc->x1 || y < rc->y0 || y > rc->y1)) ||
        (rc->axis == YZ && (z < rc->z0 || z > rc->z1 || y < rc->y0 || y > rc->y1))) return 0;
    rec->u = rc->axis == YZ ? (y - rc->y0) / (rc->y1 - rc->y0) : (x - rc->x0) / (rc->x1 - rc->x0);
    rec->v = rc->axis == XY ? (y - rc->y0) / (rc->y1 - rc->y0) : (z - rc->z0) / (rc->z1 - rc->z0);
    rec->t = t;
    set_facenormal(rec, r, V(rc->axis == YZ ? 1 : 0, rc->axis == XZ ? 1 : 0, rc->axis == XY ? 1 : 0));
    rec->mat = rc->mat; rec->p = ray_at(r, t);
    CNT(rect_hits);
    return 1;
}
static int medium_hit(constant_medium *m, ray *r, double tmin, double tmax, hit_record *rec) { /* :563-584 */
    CNT(medium_tests);
    hit_record r1, r2;
    if (!hittable_hit(m->boundary, r, -HUGE_VAL, HUGE_VAL, &r1)) return 0;
    if (!hittable_hit(m->boundary, r, r1.t + 0.0001, HUGE_VAL, &r2)) return 0;
    if (r1.t < tmin) r1.t = tmin;
    if (r2.t > tmax) r2.t = tmax;
    if (r1.t >= r2.t) return 0;
    if (r1.t < 0) r1.t = 0;
    const double ray_length = vlen(r->dir);
    const double inside = (r2.t - r1.t) * ray_length;
    const double hit_distance = m->neg_inv_density * log(random_double());
    if (hit_distance > inside) return 0;
    rec->t = r1.t + hit_distance / ray_length; rec->p = ray_at(r, rec->t);
    rec->normal = V(1, 0, 0); rec->front_face = 1; rec->mat = m->phase_function;
    /* u,v are left unset by the reference (rec is caller stack garbage); the oracle zeroes them */
    rec->u = 0; rec->v = 0;
    return 1;
}
int hittable_hit(hittable *o, ray *r, double tmin, double tmax, hit_record *rec) { /* :595-615 */
    switch (o->t) {
    case HITTABLE_SPHERE: return sphere_hit(o->obj, r, tmin, tmax, rec);
    case HITTABLE_MOVING_SPHERE: { moving_sphere *s = o->obj; CNT(msphere_tests); return sphere_hit_center(msphere_center(s, r->time), s->r, s->mat, r, tmin, tmax, rec); } /* :462-481 */
    case HITTABLE_BVH_NODE: { /* :485-490 */
        bvh_node *n = o->obj;
        if (aabb_hit(&n->box, r, tmin, tmax) == 0) return 0;
        int hl = hittable_hit(n->left, r, tmin, tmax, rec);
        int hr = hittable_hit(n->right, r, tmin, hl ? rec->t : tmax, rec);
        return hl || hr;
    }
    case HITTABLE_RECT: return rect_hit(o->obj, r, tmin, tmax, rec);
    case HITTABLE_BOX: return hittable_list_hit(((box *)o->obj)->sides, r, tmin, tmax, rec);
    case HITTABLE_TRANSLATE: { /* :527-536 */
        translate *b = o->obj; CNT(translate_calls);
        ray moved = {vsub(r->orig, b->offset), r->dir, r->time};
        if (!hittable_hit(b->obj, &moved, tmin, tmax, rec)) return 0;
        CNT(translate_hits);
        rec->p = vadd(rec->p, b->offset);
        set_facenormal(rec, &moved, rec->normal);
        return 1;
    }
    case HITTABLE_ROTATE: { /* :540-559; face normal tested against the OBJECT-space direction (F11) */
        rotate *ob = o->obj; CNT(rotate_calls);
        point3 origin = r->orig; vec3 direction = r->dir;
        rotatePoint(r->orig, -ob->sin_theta, ob->cos_theta, ob->axis, &origin);
        rotatePoint(r->dir, -ob->sin_theta, ob->cos_theta, ob->axis, &direction);
        ray rr = {origin, direction, r->time};
        if (!hittable_hit(ob->obj, &rr, tmin, tmax, rec)) return 0;
        CNT(rotate_hits);
        vec3 p = rec->p, n = rec->normal;
        rotatePoint(rec->p, ob->sin_theta, ob->cos_theta, ob->axis, &p);
        rotatePoint(rec->normal, ob->sin_theta, ob->cos_theta, ob->axis, &n);
        rec->p = p; set_facenormal(rec, &rr, n);
        return 1;
    }
    case HITTABLE_CONSTANT_MEDIUM: return medium_hit(o->obj, r, tmin, tmax, rec);
    case HITTABLE_FLIP_FACE: /* :587-591: toggles front_face only */
        if (!hittable_hit(((flip_face *)o->obj)->obj, r, tmin, tmax, rec)) return 0;
        rec->front_face = !rec->front_face; return 1;
    }
    printf("!! Not implemented hittable_hit for type %d\n", o->t); return 0;
}

/* bounding boxes: hittable.c:638-737 */
int hittable_bounding_box(hittable *o, double t0, double t1, aabb *out) {
    switch (o->t) {
    case HITTABLE_SPHERE: { sphere *s = o->obj; vec3 rr = V(s->r, s->r, s->r); out->minimum = vsub(s->center, rr); out->maximum = vadd(s->center, rr); return 1; }
    case HITTABLE_MOVING_SPHERE: {
        moving_sphere *s = o->obj; vec3 rr = V(s->r, s->r, s->r);
        aabb b0 = {vsub(msphere_center(s, t0), rr), vadd(msphere_center(s, t0), rr)}, b1 = {vsub(msphere_center(s, t1), rr), vadd(msphere_center(s, t1), rr)};
        *out = surrounding_box(b0, b1); return 1;
    }
    case HITTABLE_BVH_NODE: *out = ((bvh_node *)o->obj)->box; return 1;
    case HITTABLE_RECT: {
        rect *rc = o->obj;
        if (rc->axis == XY) { out->minimum = V(rc->x0, rc->y0, rc->k - 0.0001); out->maximum = V(rc->x1, rc->y1, rc->k + 0.0001); }
        else if (rc->axis == XZ) { out->minimum = V(rc->x0, rc->k - 0.0001, rc->z0); out->maximum = V(rc->x1, rc->k + 0.0001, rc->z1); }
        else { out->minimum = V(rc->k - 0.0001, rc->y0, rc->z0); out->maximum = V(rc->k + 0.0001, rc->y1, rc->z1); }
        return 1;
    }
    case HITTABLE_BOX: { box *b = o->obj; out->minimum = b->box_min; out->maximum = b->box_max; return 1; }
    case HITTABLE_TRANSLATE: {
        translate *t = o->obj;
        if (!hittable_bounding_box(t->obj, t0, t1, out)) return 0;
        out->minimum = vadd(out->minimum, t->offset); out->maximum = vadd(out->maximum, t->offset); return 1;
    }
    case HITTABLE_ROTATE: { rotate *r = o->obj; *out = r->bbox; return r->hasbox; }
    case HITTABLE_CONSTANT_MEDIUM: return hittable_bounding_box(((constant_medium *)o->obj)->boundary, t0, t1, out);
    case HITTABLE_FLIP_FACE: return hittable_bounding_box(((flip_face *)o->obj)->obj, t0, t1, out);
    }
    printf("!! Not implemented hittable_hit for type %d\n", o->t); return 0;
}

/* light pdfs and sampling: hittable.c:760-864 */
double hittable_pdf_value(hittable *o, point3 orig, vec3 v) {
    if (o->t == HITTABLE_SPHERE) { /* :760-768, tmin 0.001 */
        sphere *s = o->obj; hit_record rec; ray r = {orig, v, 0};
        CNT(light_pdf_sphere);
        if (!sphere_hit(s, &r, 0.001, HUGE_VAL, &rec)) return 0;
        CNT(light_pdf_sphere_hits);
        double cos_theta_max = sqrt(1 - s->r * s->r / vlen2(vsub(s->center, orig)));
        double solid_angle = 2 * PI * (1 - cos_theta_max);
        return 1 / solid_angle;
    } else if (o->t == HITTABLE_RECT) { /* :772-788, tmin 0.0001 */
        rect *rc = o->obj; hit_record rec; ray r = {orig, v, 0};
        CNT(light_pdf_rect);
        if (!rect_hit(rc, &r, 0.0001, HUGE_VAL, &rec)) return 0;
        CNT(light_pdf_rect_hits);
        double distance_squared = rec.t * rec.t * vlen2(v);
        double area = 0;
        if (rc->axis == XZ) area = (rc->x1 - rc->x0) * (rc->z1 - rc->z0);
        else if (rc->axis == XY) area = (rc->x1 - rc->x0) * (rc->y1 - rc->y0);
        else if (rc->axis == YZ) area = (rc->y1 - rc->y0) * (rc->z1 - rc->z0);
        double cosine = fabs(vdot(v, rec.normal) / vlen(v));
        return distance_squared / (cosine * area);
    } else if (o->t == HITTABLE_FLIP_FACE) return hittable_pdf_value(((flip_face *)o->obj)->obj, orig, v);
    return 0.0;
}
vec3 hittable_random(hittable *o, point3 orig) {
    if (o->t == HITTABLE_SPHERE) { /* :821-835 */
        sphere *s = o->obj;
        CNT(light_sample_sphere);
        vec3 direction = vsub(s->center, orig);
        double distance_squared = vlen2(direction);
        onb uvw; onb_init_from_w(&uvw, &direction);
        double r1 = random_double(), r2 = random_double();
        double z = 1 + r2 * (sqrt(1 - s->r * s->r / distance_squared) - 1);
        double phi = 2 * PI * r1;
        double x = cos(phi) * sqrt(1 - z * z), y = sin(phi) * sqrt(1 - z * z);
        return onb_localv(&uvw, V(x, y, z));
    } else if (o->t == HITTABLE_RECT) { /* :838-848 */
        rect *rc = o->obj; point3 p = {0, 0, 0};
        CNT(light_sample_rect);
        if (rc->axis == XZ) { double a = random_double_scaled(rc->x0, rc->x1), b = random_double_scaled(rc->z0, rc->z1); p = V(a, rc->k, b); }
        else if (rc->axis == XY) { double a = random_double_scaled(rc->x0, rc->x1), b = random_double_scaled(rc->y0, rc->y1); p = V(a, b, rc->k); }
        else if (rc->axis == YZ) { double a = random_double_scaled(rc->y0, rc->y1), b = random_double_scaled(rc->z0, rc->z1); p = V(rc->k, a, b); }
        return vsub(p, orig);
    } else if (o->t == HITTABLE_FLIP_FACE) return hittable_random(((flip_face *)o->obj)->obj, orig);
    return V(1, 0, 0);
}

/* ------------------------------------------------------------------ */
/* pdf: pdf.c:12-198                                                   */
/* ------------------------------------------------------------------ */
pdf *pdf_cosine_init(vec3 *w) { pdf_cosine *c = malloc(sizeof *c); onb_init_from_w(&c->uvw, w); pdf *p = malloc(sizeof *p); p->type = PDF_COSINE; p->obj = c; return p; }
void pdf_hittable_list_init_stack(pdf *wrapper, pdf_hittable_list *inst, hittable_list *obj, point3 origin) { inst->obj = obj; inst->orig = origin; wrapper->obj = inst; wrapper->type = PDF_HITTABLE_LIST; }
void pdf_mixture_init_stack(pdf *wrapper, pdf_mixture *inst, pdf *a, pdf *b, double ratio) { inst->a = a; inst->b = b; inst->ratio = ratio; wrapper->type = PDF_MIXTURE; wrapper->obj = inst; }
void pdf_free(pdf *p) { free(p->obj); free(p); }
double pdf_cosine_value(pdf_cosine *p, vec3 *dir) { double cosine = vdot(vunit(*dir), p->uvw.axis[2]); return (cosine <= 0) ? 0 : cosine / PI; } /* :100-103 */
double pdf_value(pdf *p, vec3 *dir) { /* :125-137 */
    switch (p->type) {
    case PDF_COSINE: return pdf_cosine_value(p->obj, dir);
    case PDF_HITTABLE: { pdf_hittable *h = p->obj; return hittable_pdf_value(h->obj, h->orig, *dir); }
    case PDF_HITTABLE_LIST: { pdf_hittable_list *h = p->obj; return hittable_list_pdf_value(h->obj, &h->orig, dir); }
    case PDF_MIXTURE: { pdf_mixture *m = p->obj; return m->ratio * pdf_value(m->a, dir) + (1 - m->ratio) * pdf_value(m->b, dir); }
    }
    return 0;
}
vec3 pdf_generate(pdf *p) { /* :185-198 */
    switch (p->type) {
    case PDF_COSINE: CNT(cosine_sample); return onb_localv(&((pdf_cosine *)p->obj)->uvw, random_cosine_direction());
    case PDF_HITTABLE: { pdf_hittable *h = p->obj; return hittable_random(h->obj, h->orig); }
    case PDF_HITTABLE_LIST: { pdf_hittable_list *h = p->obj; return hittable_list_random(h->obj, h->orig); }
    case PDF_MIXTURE: { pdf_mixture *m = p->obj; if (random_double() < m->ratio) return pdf_generate(m->a); return pdf_generate(m->b); }
    }
    return V(0, 0, 0);
}

/* ------------------------------------------------------------------ */
/* camera: camera.c:5-53 (stage 1: 01 camera.h:30-82 has no time draw)  */
/* ------------------------------------------------------------------ */
camera *camera_new(point3 lookfrom, point3 lookat, vec3 vup, double vfov, double aspect_ratio, double aperture, double focus_dist, double time0, double time1) {
    camera *c = malloc(sizeof *c);
    double theta = deg2rad(vfov);
    double h = tan(theta / 2.0);
    double viewport_height = 2.0 * h, viewport_width = aspect_ratio * viewport_height;
    c->w = vunit(vsub(lookfrom, lookat));
    c->u = vunit(vcross(vup, c->w));
    c->v = vcross(c->w, c->u);
    c->origin = lookfrom;
    c->horizontal = vscale(vscale(c->u, viewport_width), focus_dist);
    c->vertical = vscale(vscale(c->v, viewport_height), focus_dist);
    c->lower_left_corner = vsub(vsub(vsub(c->origin, vdiv(c->horizontal, 2.0)), vdiv(c->vertical, 2.0)), vscale(c->w, focus_dist));
    c->lens_radius = aperture / 2.0; c->time0 = time0; c->time1 = time1;
    return c;
}
void camera_free(camera *c) { free(c); }
ray camera_get_ray(camera *c, double s, double t) {
    CNT(camera_rays);
    vec3 rd = vscale(random_in_unit_disk(), c->lens_radius);
    vec3 offset = vadd(vscale(c->u, rd.x), vscale(c->v, rd.y));
    vec3 dir = vsub(vadd(vadd(c->lower_left_corner, vscale(c->horizontal, s)), vscale(c->vertical, t)), c->origin);
    dir = vsub(dir, offset);
    ray r = {vadd(c->origin, offset), dir, 0.0};
    if (g_stage != 1) r.time = random_double_scaled(c->time0, c->time1);
    return r;
}

/* ------------------------------------------------------------------ */
/* integrators                                                         */
/* ------------------------------------------------------------------ */
void write_color(unsigned char *pixels, color *pc, int spp, int index) { /* ray.c:35-47 */
    double scale = 1.0 / (double)spp;
    double r = sqrt(scale * pc->x), g = sqrt(scale * pc->y), b = sqrt(scale * pc->z);
    pixels[index] = (unsigned char)(256 * clamp(r, 0.0, 0.999));
    pixels[index + 1] = (unsigned char)(256 * clamp(g, 0.0, 0.999));
    pixels[index + 2] = (unsigned char)(256 * clamp(b, 0.0, 0.999));
}
static color sky(ray *r) { /* 01 main.c:59-64 */
    vec3 ud = vunit(r->dir); double t = 0.5 * (ud.y + 1.0);
    return vadd(vscale(V(1, 1, 1), (1.0 - t)), vscale(V(0.5, 0.7, 1.0), t));
}
/* stage-3 recursion, ray.c:50-84 */
static color rc3(ray *r, color *bg, hittable_list *world, hittable_list *lights, int depth, double lw) {
    if (depth <= 0) return V(0, 0, 0);
    hit_record rec; CNT(rays);
    if (hittable_list_hit(world, r, 0.001, HUGE_VAL, &rec) == 0) { CNT(miss); return *bg; }
    scatter_record srec;
    color emitted = material_emitted(rec.mat, &rec, rec.u, rec.v, rec.p);
    if (material_scatter(rec.mat, r, &rec, &srec) == 0) { CNT(emit); return emitted; }
    if (srec.is_specular) {
        if (rec.mat->type == MATERIAL_METAL) CNT(specular_metal); else CNT(specular_dielectric);
        return vmul(srec.attenuation, rc3(&srec.specular_ray, bg, world, lights, depth - 1, lw));
    }
    CNT(diffuse);
    pdf light_pdf; pdf_hittable_list light_inst; pdf_hittable_list_init_stack(&light_pdf, &light_inst, lights, rec.p);
    pdf mix; pdf_mixture mix_inst; pdf_mixture_init_stack(&mix, &mix_inst, &light_pdf, srec.pdf_ptr, lw);
    ray scattered = {rec.p, pdf_generate(&mix), r->time};
    double pdf_val = pdf_value(&mix, &scattered.dir);
    if (srec.pdf_ptr != 0) pdf_free(srec.pdf_ptr);
    color a = vscale(srec.attenuation, material_scattering_pdf(rec.mat, r, &rec, &scattered));
    color child = rc3(&scattered, bg, world, lights, depth - 1, lw);
    color w = vdiv(child, pdf_val);
    return vadd(emitted, vmul(a, w));
}
color ray_color(ray *r, color *background, hittable_list *world, hittable_list *lights, int depth) { return rc3(r, background, world, lights, depth, 0.1); }
/* stage-2 (02 main.c:44-63) and stage-1 (01 main.c:39-65) recursion */
static color rc21(ray *r, const orc_settings *st, hittable_list *world, int depth) {
    if (depth <= 0) return V(0, 0, 0);
    hit_record rec; CNT(rays);
    if (hittable_list_hit(world, r, 0.001, HUGE_VAL, &rec) == 0) { CNT(miss); return st->integrator == 2 ? sky(r) : st->background; }
    ray scattered; color att;
    color emitted = st->integrator == 2 ? V(0, 0, 0) : material_emitted(rec.mat, &rec, rec.u, rec.v, rec.p);
    if (material_scatter_att(rec.mat, r, &rec, &att, &scattered) == 0) { CNT(emit); return emitted; }
    CNT(diffuse);
    color child = rc21(&scattered, st, world, depth - 1);
    if (st->integrator == 2) return vmul(att, child);
    return vadd(emitted, vmul(att, child));
}
color orc_ray_color(ray *r, const orc_settings *st, hittable_list *world, hittable_list *lights, int depth) {
    color bg = st->background;
    if (st->integrator == 0) return rc3(r, &bg, world, lights, depth, st->mixture_light_weight);
    return rc21(r, st, world, depth);
}
/* Loop form (what the GPU executes): L += T*emitted; T *= weight; stop on miss / no scatter / depth. */
color orc_ray_color_iter(ray *r0, const orc_settings *st, hittable_list *world, hittable_list *lights, uint64_t *rays_out) {
    color L = V(0, 0, 0), T = V(1, 1, 1); ray r = *r0; uint64_t nr = 0;
    for (int b = 0; b < st->max_depth; ++b) {
        hit_record rec; ++nr; CNT(rays);
        if (hittable_list_hit(world, &r, 0.001, HUGE_VAL, &rec) == 0) {
            CNT(miss);
            color bg = st->integrator == 2 ? sky(&r) : st->background;
            L = vadd(L, vmul(T, bg)); break;
        }
        if (st->integrator == 0) {
            scatter_record srec;
            color emitted = material_emitted(rec.mat, &rec, rec.u, rec.v, rec.p);
            if (material_scatter(rec.mat, &r, &rec, &srec) == 0) { CNT(emit); L = vadd(L, vmul(T, emitted)); break; }
            if (srec.is_specular) {
                if (rec.mat->type == MATERIAL_METAL) CNT(specular_metal); else CNT(specular_dielectric);
                T = vmul(T, srec.attenuation); r = srec.specular_ray; continue;
            }
            CNT(diffuse);
            L = vadd(L, vmul(T, emitted));
            pdf light_pdf; pdf_hittable_list li; pdf_hittable_list_init_stack(&light_pdf, &li, lights, rec.p);
            pdf mix; pdf_mixture mi; pdf_mixture_init_stack(&mix, &mi, &light_pdf, srec.pdf_ptr, st->mixture_light_weight);
            ray scattered = {rec.p, pdf_generate(&mix), r.time};
            double pdf_val = pdf_value(&mix, &scattered.dir);
            if (srec.pdf_ptr != 0) pdf_free(srec.pdf_ptr);
            double spdf = material_scattering_pdf(rec.mat, &r, &rec, &scattered);
            T = vmul(T, vdiv(vscale(srec.attenuation, spdf), pdf_val));
            r = scattered;
        } else {
            ray scattered; color att;
            color emitted = st->integrator == 2 ? V(0, 0, 0) : material_emitted(rec.mat, &rec, rec.u, rec.v, rec.p);
            L = vadd(L, vmul(T, emitted));
            if (material_scatter_att(rec.mat, &r, &rec, &att, &scattered) == 0) { CNT(emit); break; }
            T = vmul(T, att); r = scattered;
        }
    }
    if (rays_out) *rays_out += nr;
    return L;
}

/* ------------------------------------------------------------------ */
/* renders                                                             */
/* ------------------------------------------------------------------ */
static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}
int orc_render(hittable_list *world, hittable_list *lights, camera *cam, const orc_settings *st,
               int W, int H, int s_begin, int s_end, int rng_mode, uint64_t seed,
               int threads, int iterative, double *rgb_sum, orc_render_stats *stats) {
    if (rng_mode == ORC_RNG_GLIBC) threads = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    uint64_t total_rays = 0;
    double t0 = now_s();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : total_rays)
    for (int j = 0; j < H; ++j) {
        orc_rng_mode(rng_mode);
        for (int i = 0; i < W; ++i) {
            color acc = V(0, 0, 0);
            for (int s = s_begin; s < s_end; ++s) {
                if (rng_mode == ORC_RNG_PHILOX) orc_rng_seek(seed, (uint32_t)(j * W + i), (uint32_t)s);
                double u = ((double)i + random_double()) / (double)(W - 1); /* main.c:150-151 */
                double v = ((double)j + random_double()) / (double)(H - 1);
                ray r = camera_get_ray(cam, u, v);
                CNT(samples);
                color c;
                if (iterative) c = orc_ray_color_iter(&r, st, world, lights, &total_rays);
                else c = orc_ray_color(&r, st, world, lights, st->max_depth);
                if (st->scrub_nan && !(isfinite(c.x) && isfinite(c.y) && isfinite(c.z))) c = V(0, 0, 0);
                acc = vadd(acc, c);
            }
            double *o = rgb_sum + 3 * ((size_t)j * W + i);
            o[0] += acc.x; o[1] += acc.y; o[2] += acc.z;
        }
        counters_flush();
    }
    if (stats) { stats->samples = (uint64_t)W * H * (uint64_t)(s_end - s_begin); stats->rays = total_rays; stats->seconds = now_s() - t0; }
    return 0;
}
void orc_render_refloop_u8(hittable_list *world, hittable_list *lights, camera *cam, const orc_settings *st,
                           int W, int H, int spp, unsigned char *pixels) { /* main.c:132-160, one thread */
    const unsigned THREADS = 16;
    int step = (int)floor(H / THREADS);
    orc_rng_mode(ORC_RNG_GLIBC);
    for (unsigned k = 0; k < THREADS; ++k)
        for (int sj = step - 1; sj >= 0; --sj) {
            int j = sj + (int)(k * step);
            for (int i = 0; i < W; ++i) {
                color pc = V(0, 0, 0);
                for (int s = 0; s < spp; ++s) {
                    double u = ((double)i + random_double()) / (double)(W - 1);
                    double v = ((double)j + random_double()) / (double)(H - 1);
                    ray r = camera_get_ray(cam, u, v);
                    pc = vadd(pc, orc_ray_color(&r, st, world, lights, st->max_depth));
                }
                write_color(pixels, &pc, spp, (i + (j * W)) * 3);
            }
        }
}
/* 01 main.c:30-35: average, gamma 2, clamp, x256, printed as decimal text */
void orc_write_color_ppm(FILE *f, color *pc, int spp) {
    double scale = 1.0 / (double)spp;
    double r = sqrt(scale * pc->x), g = sqrt(scale * pc->y), b = sqrt(scale * pc->z);
    fprintf(f, "%d %d %d\n", (int)(256 * clamp(r, 0.0, 0.999)), (int)(256 * clamp(g, 0.0, 0.999)), (int)(256 * clamp(b, 0.0, 0.999)));
}
/* 01 main.c:168-191: header, then rows from the top, samples in order, the rand() stream is NOT reseeded */
int orc_render_ppm01(hittable_list *world, camera *cam, const orc_settings *st, int W, int H, int spp, const char *path) {
    FILE *f = fopen(path, "w+");
    if (!f) return -1;
    orc_rng_mode(ORC_RNG_GLIBC);
    fprintf(f, "P3\n%d %d\n255\n", W, H);
    for (int j = H - 1; j >= 0; --j)
        for (int i = 0; i < W; ++i) {
            color pc = V(0, 0, 0);
            for (int s = 0; s < spp; ++s) {
                double u = ((double)i + random_double()) / (double)(W - 1);
                double v = ((double)j + random_double()) / (double)(H - 1);
                ray r = camera_get_ray(cam, u, v);
                pc = vadd(pc, orc_ray_color(&r, st, world, NULL, st->max_depth));
            }
            orc_write_color_ppm(f, &pc, spp);
        }
    fclose(f);
    return 0;
}
