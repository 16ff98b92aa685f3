/*
 * scene_api.c -- host implementation of include/raytracer.h (product code, plain C).
 *
 * Behaviour follows the reference's constructors (paths under /root/reference/03_FullTracer):
 * what is stored, which list an object joins, what is precomputed on the host (rotation sin/cos and
 * rotated bounds hittable.c:246-288, camera basis camera.c:5-38, perlin tables perlin.c:13-29,67-75) and
 * how many rand() draws construction consumes.  Host precomputation is done in double with the
 * reference's operation order because these values are shipped verbatim to the FP64 verify kernel.
 */
#include "scene_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- small helpers ------------------------------------------------------------------------- */
static vec3 v3(double x, double y, double z) { vec3 v = {x, y, z}; return v; }
static vec3 add3(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static vec3 sub3(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static vec3 mul3(vec3 a, double k) { return v3(a.x * k, a.y * k, a.z * k); }
static vec3 div3(vec3 a, double k) { return v3(a.x / k, a.y / k, a.z / k); }
static double len3(vec3 a) { return sqrt((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); }
static vec3 cross3(vec3 a, vec3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static vec3 unit3(vec3 a) { return div3(a, len3(a)); }

typedef struct { void **v; int n, cap; } registry;
static void reg_push(registry *r, void *p) {
    if (r->n == r->cap) { r->cap = r->cap ? 2 * r->cap : 256; r->v = (void **)realloc(r->v, sizeof(void *) * (size_t)r->cap); }
    r->v[r->n++] = p;
}
static registry g_textures, g_materials, g_lists;
/* Bumped by every constructor and every free: ray_color (rtc_api.cpp) keeps the device tables of the last scene it was handed and must
 * not trust a (world, lights) pointer pair whose lists have been freed and rebuilt at the same addresses in between. */
static unsigned long long g_generation;
unsigned long long rt_api_generation(void) { return g_generation; }

/* ---- utils.h:21-26 -------------------------------------------------------------------------- */
double random_double(void) { return rand() / (RAND_MAX + 1.0); }
double random_double_scaled(double min, double max) { return min + (max - min) * random_double(); }
int random_int(int min, int max) { return (int)random_double_scaled(min, max + 1); }
double deg2rad(double deg) { return deg * PI / 180.0; }

/* ---- textures ------------------------------------------------------------------------------- */
static texture *tex_alloc(rt_tex_kind kind) { ++g_generation;
    texture *t = (texture *)calloc(1, sizeof *t);
    t->kind = kind; t->id = g_textures.n; reg_push(&g_textures, t); return t;
}
texture *texture_solid_color_init(color c) { texture *t = tex_alloc(RT_TEX_SOLID); t->value = c; return t; }
texture *texture_solid_color_init_rgb(double r, double g, double b) { return texture_solid_color_init(v3(r, g, b)); }
texture *texture_checker_init(texture *odd, texture *even) { texture *t = tex_alloc(RT_TEX_CHECKER); t->odd = odd; t->even = even; return t; }
texture *texture_checker_init_c(color odd, color even) {
    texture *o = texture_solid_color_init(odd); texture *e = texture_solid_color_init(even); return texture_checker_init(o, e);
}
static void shuffle_perm(int *p) { /* perlin.c:13-29: identity then swap i with random_int(0,i) */
    for (int i = 0; i < 256; ++i) p[i] = i;
    for (int i = 0; i < 256; ++i) { int j = random_int(0, i); int tmp = p[i]; p[i] = p[j]; p[j] = tmp; }
}
texture *texture_noise_init_scaled(double sc) {
    texture *t = tex_alloc(RT_TEX_NOISE); t->scale = sc;
    t->ranvec = (double *)malloc(sizeof(double) * 768); t->perm = (int *)malloc(sizeof(int) * 768);
    for (int i = 0; i < 256; ++i) { /* perlin.c:70: unit(random_scaled(-1,1)^3), draws left to right */
        double a = random_double_scaled(-1, 1), b = random_double_scaled(-1, 1), c = random_double_scaled(-1, 1);
        vec3 u = unit3(v3(a, b, c)); t->ranvec[3 * i] = u.x; t->ranvec[3 * i + 1] = u.y; t->ranvec[3 * i + 2] = u.z;
    }
    shuffle_perm(t->perm); shuffle_perm(t->perm + 256); shuffle_perm(t->perm + 512);
    return t;
}
texture *texture_noise_init(void) { return texture_noise_init_scaled(1.0); }
texture *texture_image_init_rgb(const unsigned char *rgb, int width, int height) {
    texture *t = tex_alloc(RT_TEX_IMAGE); t->width = width; t->height = height;
    t->pixels = (unsigned char *)malloc((size_t)width * height * 3); memcpy(t->pixels, rgb, (size_t)width * height * 3);
    return t;
}
/* "P6 <width> <height> <maxval><one whitespace byte>" read from a buffer of n bytes that is NOT NUL-terminated (a file image):
 * a bounded scanner instead of sscanf, which runs strlen over its input first.  '#' comments run to the end of the line. */
static int ppm_header(const unsigned char *p, size_t n, int *w, int *h, int *maxval, size_t *data_offset) {
    size_t i = 2;
    int *field[3] = {w, h, maxval};
    for (int k = 0; k < 3; k++) {
        for (;;) {   /* whitespace and comments before the number */
            while (i < n && (p[i] == ' ' || p[i] == '\t' || p[i] == '\n' || p[i] == '\r' || p[i] == '\v' || p[i] == '\f')) i++;
            if (i < n && p[i] == '#') { while (i < n && p[i] != '\n') i++; continue; }
            break;
        }
        long v = 0; int digits = 0;
        while (i < n && p[i] >= '0' && p[i] <= '9' && digits < 9) { v = v * 10 + (p[i] - '0'); i++; digits++; }
        if (digits == 0 || (i < n && p[i] >= '0' && p[i] <= '9')) return 0;   /* no number, or more than 9 digits */
        *field[k] = (int)v;
    }
    if (i >= n || !(p[i] == ' ' || p[i] == '\t' || p[i] == '\n' || p[i] == '\r')) return 0;   /* the single separator before the samples */
    *data_offset = i + 1;
    return 1;
}
/* texture.c:63-70 loads through stb_image, which takes the format from the file's first bytes; here: JPEG, baseline and progressive
 * (jpeg_decode.c), PNG and BMP (image_decode.c), binary PGM / PPM (P5 / P6, maxval <= 255, bytes as they are).  The formats stb also
 * reads and no scene of the reference uses -- TGA, PSD, GIF, HDR, PIC -- are refused.  NULL + the reference's message on failure. */
texture *texture_image_init(char *filename) {
    texture *t = NULL;
    unsigned char *file = NULL, *rgb = NULL;
    long size = 0; int w = 0, h = 0;
    FILE *f = filename ? fopen(filename, "rb") : NULL;
    if (f && fseek(f, 0, SEEK_END) == 0 && (size = ftell(f)) > 0 && fseek(f, 0, SEEK_SET) == 0 && (file = (unsigned char *)malloc((size_t)size)) != NULL
        && fread(file, 1, (size_t)size, f) == (size_t)size) {
        if (size > 3 && file[0] == 0xFF && file[1] == 0xD8) {
            if (rt_jpeg_decode(file, (size_t)size, &w, &h, &rgb)) t = texture_image_init_rgb(rgb, w, h);
        } else if (size > 8 && file[0] == 137 && file[1] == 'P') {
            if (rt_png_decode(file, (size_t)size, &w, &h, &rgb)) t = texture_image_init_rgb(rgb, w, h);
        } else if (size > 26 && file[0] == 'B' && file[1] == 'M') {
            if (rt_bmp_decode(file, (size_t)size, &w, &h, &rgb)) t = texture_image_init_rgb(rgb, w, h);
        } else if (size > 8 && file[0] == 'P' && (file[1] == '6' || file[1] == '5')) {
            int mx = 0; size_t data = 0;
            const size_t chan = file[1] == '6' ? 3 : 1;
            if (ppm_header(file, (size_t)size, &w, &h, &mx, &data) && mx > 0 && mx <= 255 && w > 0 && h > 0 && (long long)w * h <= (1ll << 28)
                && (size_t)w * (size_t)h * chan <= (size_t)size - data) {
                if (chan == 3) t = texture_image_init_rgb(file + data, w, h);
                else if ((rgb = (unsigned char *)malloc((size_t)w * h * 3)) != NULL) {
                    for (size_t i = 0; i < (size_t)w * h; i++) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = file[data + i];
                    t = texture_image_init_rgb(rgb, w, h);
                }
            }
        }
    }
    if (f) fclose(f);
    free(file); free(rgb);
    if (!t) { printf("!! could not load texture image file '%s'! \n", filename ? filename : "(null)"); fflush(stdout); }
    return t;
}
/* texture.c:76-90 frees one texture and leaves its pointer in the registry (a later textures_free_all would free it twice);
 * here the slot is cleared.  A checker does not own its two sub-textures: they are registered on their own. */
void texture_free(texture *t) { ++g_generation;
    if (!t) return;
    if (t->id >= 0 && t->id < g_textures.n && g_textures.v[t->id] == t) g_textures.v[t->id] = NULL;
    free(t->ranvec); free(t->perm); free(t->pixels); free(t);
}
void textures_free_all(void) { ++g_generation;
    for (int i = 0; i < g_textures.n; i++) { texture *t = (texture *)g_textures.v[i]; if (!t) continue; free(t->ranvec); free(t->perm); free(t->pixels); free(t); }
    g_textures.n = 0;
}

/* ---- materials ------------------------------------------------------------------------------ */
static material *mat_alloc(rt_mat_kind kind) { ++g_generation;
    material *m = (material *)calloc(1, sizeof *m); m->kind = kind; m->id = g_materials.n; reg_push(&g_materials, m); return m;
}
material *material_lambertian_new(texture *t) { material *m = mat_alloc(RT_MAT_LAMBERTIAN); m->tex = t; return m; }
material *material_lambertian_new_from_color(color c) { return material_lambertian_new(texture_solid_color_init(c)); }
material *material_metal_new(color a, double fuzz) { material *m = mat_alloc(RT_MAT_METAL); m->albedo = a; m->param = fuzz; return m; }
material *material_dielectric_new(double ir) { material *m = mat_alloc(RT_MAT_DIELECTRIC); m->param = ir; return m; }
material *material_light_new(texture *t) { material *m = mat_alloc(RT_MAT_LIGHT); m->tex = t; return m; }
material *material_light_new_from_color(color c) { return material_light_new(texture_solid_color_init(c)); }
material *material_isotropic_new(texture *t) { material *m = mat_alloc(RT_MAT_ISOTROPIC); m->tex = t; return m; }
material *material_isotropic_new_from_color(color c) { return material_isotropic_new(texture_solid_color_init(c)); }
/* material.c:26-33: one material (its texture stays with the texture registry); the registry slot is cleared */
void material_free(material *m) { ++g_generation;
    if (!m) return;
    if (m->id >= 0 && m->id < g_materials.n && g_materials.v[m->id] == m) g_materials.v[m->id] = NULL;
    free(m);
}
void materials_free_all(void) { ++g_generation; for (int i = 0; i < g_materials.n; i++) free(g_materials.v[i]); g_materials.n = 0; }

/* ---- lists ---------------------------------------------------------------------------------- */
hittable_list *hittable_list_new(int max_size) { ++g_generation;
    hittable_list *l = (hittable_list *)calloc(1, sizeof *l);
    l->size = max_size > 4 ? max_size : 4; l->objs = (hittable **)malloc(sizeof(hittable *) * (size_t)l->size); l->gc = 1;
    reg_push(&g_lists, l); return l;
}
hittable_list *hittable_list_new_no_gc(int max_size) { hittable_list *l = hittable_list_new(max_size); l->gc = 0; return l; }
void hittable_list_add(hittable_list *l, hittable *o) { ++g_generation;
    if (l == NULL) return;
    if (l->index == l->size) { l->size *= 2; l->objs = (hittable **)realloc(l->objs, sizeof(hittable *) * (size_t)l->size); }
    l->objs[l->index++] = o;
}
int hittable_list_index(hittable_list *l) { return l->index; }
void *hittable_list_objects(hittable_list *l) { return (void *)l->objs; }
static void list_free_members(hittable_list *l) {
    if (!l->gc) return;   /* a no-gc list keeps its entries: the BVH built over it frees them (hittable_free) */
    for (int i = 0; i < l->index; i++) hittable_free(l->objs[i]);
    l->index = 0;
}
void hittable_list_free(hittable_list *l) { ++g_generation;
    if (l == NULL) return;
    for (int i = 0; i < g_lists.n; i++) if (g_lists.v[i] == l) g_lists.v[i] = NULL;   /* hittable_lists_free_all must not see it again */
    list_free_members(l);
    free(l->objs); free(l);
}
/* Members first, the lists themselves afterwards: a BVH node reads its source list while it is freed (hittable_free),
 * and that list may have been registered -- and so would have been freed -- before the list that owns the node
 * (02 main.c:232-254: the box list exists before `objects`). */
void hittable_lists_free_all(void) { ++g_generation;
    for (int i = 0; i < g_lists.n; i++) if (g_lists.v[i]) list_free_members((hittable_list *)g_lists.v[i]);
    for (int i = 0; i < g_lists.n; i++) if (g_lists.v[i]) { hittable_list *l = (hittable_list *)g_lists.v[i]; free(l->objs); free(l); }
    g_lists.n = 0;
}

/* ---- hittables ------------------------------------------------------------------------------ */
static hittable *hit_alloc(hittable_list *world, rt_hit_kind kind) { ++g_generation;
    hittable *h = (hittable *)calloc(1, sizeof *h); h->kind = kind; hittable_list_add(world, h); return h;
}
hittable *hittable_sphere_new(hittable_list *world, point3 center, double r, material *mat) {
    hittable *h = hit_alloc(world, RT_H_SPHERE); h->c0 = center; h->r = r; h->mat = mat; return h;
}
hittable *hittable_moving_sphere_new(hittable_list *world, point3 center0, point3 center1, double t0, double t1, double r, material *mat) {
    hittable *h = hit_alloc(world, RT_H_MOVING_SPHERE); h->c0 = center0; h->c1 = center1; h->t0 = t0; h->t1 = t1; h->r = r; h->mat = mat; return h;
}
hittable *hittable_rect_new(hittable_list *world, double x0, double x1, double y0, double y1, double z0, double z1, double k, rect_axis axis, material *mat) {
    hittable *h = hit_alloc(world, RT_H_RECT); h->lo = v3(x0, y0, z0); h->hi = v3(x1, y1, z1); h->k = k; h->axis = (int)axis; h->mat = mat; return h;
}
hittable *hittable_box_new(hittable_list *world, point3 mn, point3 mx, material *mat) {
    hittable *h = (hittable *)calloc(1, sizeof *h); h->kind = RT_H_BOX; h->lo = mn; h->hi = mx; h->mat = mat;
    h->members = hittable_list_new(6); /* side order of hittable.c:146-151 fixes primitive ids and tie-breaks */
    hittable_rect_new(h->members, mn.x, mx.x, mn.y, mx.y, 0, 0, mx.z, XY, mat);
    hittable_rect_new(h->members, mn.x, mx.x, mn.y, mx.y, 0, 0, mn.z, XY, mat);
    hittable_rect_new(h->members, mn.x, mx.x, 0, 0, mn.z, mx.z, mx.y, XZ, mat);
    hittable_rect_new(h->members, mn.x, mx.x, 0, 0, mn.z, mx.z, mn.y, XZ, mat);
    hittable_rect_new(h->members, 0, 0, mn.y, mx.y, mn.z, mx.z, mx.x, YZ, mat);
    hittable_rect_new(h->members, 0, 0, mn.y, mx.y, mn.z, mx.z, mn.x, YZ, mat);
    hittable_list_add(world, h);
    return h;
}
hittable *hittable_translate_init(hittable_list *world, hittable *obj, vec3 offset) {
    hittable *h = hit_alloc(world, RT_H_TRANSLATE); h->child = obj; h->offset = offset; return h;
}
static void rotate_point(vec3 o, double s, double c, int axis, vec3 *out) { /* pair choice per hittable.c:37-48 (F10) */
    if (axis == X) { out->x = c * o.x - s * o.y; out->y = s * o.x + c * o.y; }
    else if (axis == Y) { out->x = c * o.x + s * o.z; out->z = -s * o.x + c * o.z; }
    else { out->y = c * o.y - s * o.z; out->z = s * o.y + c * o.z; }
}
hittable *hittable_rotate_init(hittable_list *world, hittable *obj, double angle, rotation_axis axis) {
    hittable *h = hit_alloc(world, RT_H_ROTATE); h->child = obj; h->axis = (int)axis;
    double radians = deg2rad(angle);
    h->sin_theta = sin(radians); h->cos_theta = cos(radians);
    h->hasbox = rt_bounding_box(obj, 0, 1, &h->bbox);
    point3 mn = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, mx = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++) {
        vec3 p = v3(i * h->bbox.hi.x + (1 - i) * h->bbox.lo.x, j * h->bbox.hi.y + (1 - j) * h->bbox.lo.y, k * h->bbox.hi.z + (1 - k) * h->bbox.lo.z);
        vec3 q = p; rotate_point(p, h->sin_theta, h->cos_theta, h->axis, &q);
        mn.x = fmin(mn.x, q.x); mx.x = fmax(mx.x, q.x); mn.y = fmin(mn.y, q.y); mx.y = fmax(mx.y, q.y); mn.z = fmin(mn.z, q.z); mx.z = fmax(mx.z, q.z);
    }
    h->bbox.lo = mn; h->bbox.hi = mx;
    return h;
}
hittable *hittable_constant_medium_init(hittable_list *world, hittable *b, double d, texture *a) {
    hittable *h = hit_alloc(world, RT_H_MEDIUM); h->child = b; h->neg_inv_density = -1 / d; h->mat = material_isotropic_new(a); return h;
}
hittable *hittable_constant_medium_init_c(hittable_list *world, hittable *b, double d, color c) {
    return hittable_constant_medium_init(world, b, d, texture_solid_color_init(c));
}
hittable *hittable_flip_face_init(hittable_list *world, hittable *obj) { hittable *h = hit_alloc(world, RT_H_FLIP); h->child = obj; return h; }

/* The reference's bvh_node_constructor (hittable.c:163-215, F8) splits [start,end) of the ORIGINAL member order at
 * the midpoint, orders two-member leaves by box min.x, and draws one rand() per node.  The device BVH is built
 * separately; what has to be kept is the visit order of the members (primitive ids, tie-breaks) and the rand() count. */
static void bvh_visit_order(hittable_list *src, int start, int end, int *order, int *n) {
    (void)random_double();
    int span = end - start;
    if (span == 1) { order[(*n)++] = start; return; }
    if (span == 2) {
        rt_box a, b; int lt = 0;
        if (rt_bounding_box(src->objs[start], 0, 0, &a) && rt_bounding_box(src->objs[start + 1], 0, 0, &b)) lt = a.lo.x < b.lo.x;
        else { printf("Failed to get bounding box in bvh_node comparator.\n"); fflush(stdout); }
        order[(*n)++] = lt ? start : start + 1; order[(*n)++] = lt ? start + 1 : start;
        return;
    }
    int mid = start + span / 2;
    bvh_visit_order(src, start, mid, order, n);
    bvh_visit_order(src, mid, end, order, n);
}
hittable *bvh_node_init(hittable_list *world, hittable_list *objects, double t0, double t1) {
    hittable *h = hit_alloc(world, RT_H_BVH); h->members = objects; h->t0 = t0; h->t1 = t1;
    h->bvh_order = (int *)malloc(sizeof(int) * (size_t)(objects->index > 0 ? objects->index : 1)); h->bvh_count = 0;
    if (objects->index > 0) bvh_visit_order(objects, 0, objects->index, h->bvh_order, &h->bvh_count);
    return h;
}
void hittable_free(hittable *o) { ++g_generation; /* ownership as hittable.c:336-366; a BVH owns the members of its (no-gc) source list */
    switch (o->kind) {
    case RT_H_BVH: for (int i = 0; i < o->members->index; i++) hittable_free(o->members->objs[i]); o->members->index = 0; free(o->bvh_order); break;
    case RT_H_TRANSLATE: case RT_H_ROTATE: case RT_H_FLIP: case RT_H_MEDIUM: hittable_free(o->child); break;
    default: break; /* a box's side list is registered and freed by hittable_lists_free_all */
    }
    free(o);
}

/* bounding boxes, hittable.c:638-737 */
static rt_box box_union(rt_box a, rt_box b) {
    rt_box r = {v3(fmin(a.lo.x, b.lo.x), fmin(a.lo.y, b.lo.y), fmin(a.lo.z, b.lo.z)), v3(fmax(a.hi.x, b.hi.x), fmax(a.hi.y, b.hi.y), fmax(a.hi.z, b.hi.z))};
    return r;
}
static point3 moving_center(const hittable *s, double t) { double k = ((t - s->t0) / (s->t1 - s->t0)); return add3(mul3(sub3(s->c1, s->c0), k), s->c0); }
int rt_bounding_box(const hittable *o, double t0, double t1, rt_box *out) {
    switch (o->kind) {
    case RT_H_SPHERE: out->lo = sub3(o->c0, v3(o->r, o->r, o->r)); out->hi = add3(o->c0, v3(o->r, o->r, o->r)); return 1;
    case RT_H_MOVING_SPHERE: {
        vec3 rr = v3(o->r, o->r, o->r);
        rt_box b0 = {sub3(moving_center(o, t0), rr), add3(moving_center(o, t0), rr)}, b1 = {sub3(moving_center(o, t1), rr), add3(moving_center(o, t1), rr)};
        *out = box_union(b0, b1); return 1;
    }
    case RT_H_RECT:
        if (o->axis == XY) { out->lo = v3(o->lo.x, o->lo.y, o->k - 0.0001); out->hi = v3(o->hi.x, o->hi.y, o->k + 0.0001); }
        else if (o->axis == XZ) { out->lo = v3(o->lo.x, o->k - 0.0001, o->lo.z); out->hi = v3(o->hi.x, o->k + 0.0001, o->hi.z); }
        else { out->lo = v3(o->k - 0.0001, o->lo.y, o->lo.z); out->hi = v3(o->k + 0.0001, o->hi.y, o->hi.z); }
        return 1;
    case RT_H_BOX: out->lo = o->lo; out->hi = o->hi; return 1;
    case RT_H_BVH: {
        if (o->members->index == 0) return 0;
        rt_box acc, b;
        for (int i = 0; i < o->members->index; i++) { if (!rt_bounding_box(o->members->objs[i], t0, t1, &b)) return 0; acc = i ? box_union(acc, b) : b; }
        *out = acc; return 1;
    }
    case RT_H_TRANSLATE: if (!rt_bounding_box(o->child, t0, t1, out)) return 0; out->lo = add3(out->lo, o->offset); out->hi = add3(out->hi, o->offset); return 1;
    case RT_H_ROTATE: *out = o->bbox; return o->hasbox;
    case RT_H_MEDIUM: case RT_H_FLIP: return rt_bounding_box(o->child, t0, t1, out);
    }
    return 0;
}

/* ---- camera (camera.c:5-38) ----------------------------------------------------------------- */
camera *camera_new(point3 lookfrom, point3 lookat, vec3 vup, double vfov, double aspect_ratio, double aperture, double focus_dist, double time0, double time1) {
    camera *c = (camera *)malloc(sizeof *c);
    double theta = deg2rad(vfov);
    double h = tan(theta / 2.0);
    double viewport_height = 2.0 * h, viewport_width = aspect_ratio * viewport_height;
    c->w = unit3(sub3(lookfrom, lookat));
    c->u = unit3(cross3(vup, c->w));
    c->v = cross3(c->w, c->u);
    c->origin = lookfrom;
    c->horizontal = mul3(mul3(c->u, viewport_width), focus_dist);
    c->vertical = mul3(mul3(c->v, viewport_height), focus_dist);
    c->lower_left_corner = sub3(sub3(sub3(c->origin, div3(c->horizontal, 2.0)), div3(c->vertical, 2.0)), mul3(c->w, focus_dist));
    c->lens_radius = aperture / 2.0; c->time0 = time0; c->time1 = time1;
    return c;
}
void camera_free(camera *c) { free(c); }
void camera_fields(const camera *c, double out[24]) {
    const vec3 *v[7] = {&c->origin, &c->lower_left_corner, &c->horizontal, &c->vertical, &c->u, &c->v, &c->w};
    for (int i = 0; i < 7; i++) { out[3 * i] = v[i]->x; out[3 * i + 1] = v[i]->y; out[3 * i + 2] = v[i]->z; }
    out[21] = c->lens_radius; out[22] = c->time0; out[23] = c->time1;
}

/* camera.c:46-53, on the host, for code that wants single rays (tests, rtc_ray_colors callers): the lens sample is the
 * rejection loop of vec3.c:87-93 and the ray time one more rand() draw, both consumed even for a pinhole camera with a
 * closed shutter (F13) -- the reference's rand() stream stays in step.  The renderer does not call this: its primary rays
 * are generated on the device from Philox. */
ray camera_get_ray(camera *c, double s, double t) {
    vec3 p;
    for (;;) {   /* x first: the reference's initialiser list draws left to right */
        p.x = random_double_scaled(-1, 1); p.y = random_double_scaled(-1, 1); p.z = 0;
        if ((p.x * p.x) + (p.y * p.y) + (p.z * p.z) < 1) break;
    }
    vec3 rd = mul3(p, c->lens_radius);
    vec3 offset = add3(mul3(c->u, rd.x), mul3(c->v, rd.y));
    vec3 dir = sub3(add3(add3(c->lower_left_corner, mul3(c->horizontal, s)), mul3(c->vertical, t)), c->origin);
    ray r;
    r.dir = sub3(dir, offset); r.orig = add3(c->origin, offset); r.time = random_double_scaled(c->time0, c->time1);
    return r;
}

/* ray.c:35-47 on the host: average, gamma 2, clamp to [0, 0.999], x256 -> three bytes at `index`.  A NaN component writes
 * 0 (the reference casts it to unsigned char: undefined, F16).  rtc_tonemap_u8 is the same arithmetic for a whole image. */
void write_color(unsigned char *pixels, color *pixel_color, int samples_per_pixel, int index) {
    const double scale = 1.0 / (double)samples_per_pixel;
    const double c[3] = {pixel_color->x, pixel_color->y, pixel_color->z};
    for (int k = 0; k < 3; k++) {
        double v = sqrt(scale * c[k]);
        v = (v < 0.0) ? 0.0 : ((v > 0.999) ? 0.999 : v);
        if (!(v == v)) v = 0.0;
        pixels[index + k] = (unsigned char)(256 * v);
    }
}
