/*
 * oracle/cpu_ref.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A double-precision, plain-C restatement of the reference renderer's
 * (elkiwy/raytraCer) path-tracing hot path, written from the reference's
 * observable behaviour.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (raytracer_amd/, include/rtc.h) never links, imports or calls it.
 *
 * Parity pin: every function here is checked against the *compiled
 * reference itself* (oracle/_ref/libref03.so, libref02.so built from
 * /root/reference by oracle/Makefile) through tests/golden/make_golden.py;
 * the generated vectors live in tests/golden/.  The reference ships no
 * tests or golden vectors of its own.
 *
 * The data model deliberately mirrors the reference's public structs so
 * that one harness (oracle/harness.c) and one scene file compile unchanged
 * against the reference, against this oracle and against the product's
 * host API.  Citations are file:line under /root/reference/.
 *
 * "stage" selects which of the reference's three snapshots is restated:
 *   3 = 03_FullTracer (mixture-PDF integrator, ray.c:50-84)
 *   2 = 02_AdvancedTracer (attenuation integrator, src/main.c:44-63)
 *   1 = 01_BasicTracer (sky-gradient integrator, src/main.c:39-65)
 */
#ifndef ORACLE_CPU_REF_H
#define ORACLE_CPU_REF_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef PI
#define PI 3.1415926535897932385 /* utils.h:17 */
#endif

/* ---- math types (vec3.h:17,83-84; ray.h:21; aabb.h; onb.h) ------------- */
typedef struct { double x, y, z; } vec3;
typedef vec3 point3;
typedef vec3 color;
typedef struct { point3 orig; vec3 dir; double time; } ray;
typedef struct { point3 minimum, maximum; } aabb;
typedef struct { vec3 axis[3]; } onb;

/* ---- textures (texture.h:19-45) ---------------------------------------- */
typedef enum { TEXTURE_SOLID, TEXTURE_CHECKER, TEXTURE_NOISE, TEXTURE_IMAGE } texture_type;
typedef struct texture { texture_type type; void *o; } texture;
#define POINT_COUNT 256
typedef struct { vec3 *ranvec; int *perm_x, *perm_y, *perm_z; } perlin;
typedef struct { color color_value; } solid_color;
typedef struct { texture *odd, *even; } checker;
typedef struct { perlin *noise; double scale; } texture_noise;
typedef struct { unsigned char *data; int width, height, bytes_per_scanline; } texture_image;

/* ---- materials (material.h:18-35) -------------------------------------- */
typedef enum { MATERIAL_LAMBERTIAN, MATERIAL_METAL, MATERIAL_DIELECTRIC, MATERIAL_LIGHT, MATERIAL_ISOTROPIC } material_type;
typedef struct material { material_type type; void *mat; } material;
typedef struct { texture *albedo; } material_lambertian;
typedef struct { color albedo; double fuzz; } material_metal;
typedef struct { double ir; } material_dielectric;
typedef struct { texture *emit; } material_light;
typedef struct { texture *albedo; } material_isotropic;
struct pdf;
typedef struct scatter_record { ray specular_ray; int is_specular; color attenuation; struct pdf *pdf_ptr; } scatter_record;

/* ---- hittables (hittable.h:26-106, hittable_list.h:20-25) --------------- */
typedef struct hit_record { point3 p; vec3 normal; struct material *mat; double t; double u, v; int front_face; } hit_record;
typedef enum { HITTABLE_BVH_NODE, HITTABLE_SPHERE, HITTABLE_MOVING_SPHERE, HITTABLE_RECT, HITTABLE_BOX,
               HITTABLE_TRANSLATE, HITTABLE_ROTATE, HITTABLE_CONSTANT_MEDIUM, HITTABLE_FLIP_FACE } hittable_type;
typedef struct hittable { hittable_type t; void *obj; } hittable;
typedef struct hittable_list { struct hittable **objs; int index; int size; int gc; } hittable_list;
typedef struct { point3 center; double r; struct material *mat; } sphere;
typedef enum { XY, XZ, YZ } rect_axis;
typedef struct { struct material *mat; double x0, x1, y0, y1, z0, z1, k; rect_axis axis; } rect;
typedef struct { point3 box_min, box_max; struct hittable_list *sides; } box;
typedef struct { point3 center0, center1; double time0, time1; double r; struct material *mat; } moving_sphere;
typedef struct { hittable *left, *right; aabb box; } bvh_node;
typedef struct { hittable *boundary; struct material *phase_function; double neg_inv_density; } constant_medium;
typedef struct { hittable *obj; vec3 offset; } translate;
typedef enum { X, Y, Z } rotation_axis;
typedef struct { hittable *obj; double sin_theta, cos_theta; int hasbox; aabb bbox; rotation_axis axis; } rotate;
typedef struct { hittable *obj; } flip_face;

/* ---- pdfs (pdf.h:11-18) -------------------------------------------------- */
typedef enum { PDF_COSINE, PDF_HITTABLE, PDF_HITTABLE_LIST, PDF_MIXTURE } pdf_type;
typedef struct pdf { pdf_type type; void *obj; } pdf;
typedef struct { onb uvw; } pdf_cosine;
typedef struct { point3 orig; hittable *obj; } pdf_hittable;
typedef struct { point3 orig; hittable_list *obj; } pdf_hittable_list;
typedef struct { pdf *a, *b; double ratio; } pdf_mixture;

/* ---- camera (camera.h:17-25) --------------------------------------------- */
typedef struct { point3 origin, lower_left_corner; vec3 horizontal, vertical; vec3 u, v, w;
                 double lens_radius; double time0, time1; } camera;

/* ======================= reference-compatible API ========================= */
/* RNG front door (utils.h:23-26).  Source is switchable, see orc_rng_*. */
double random_double(void);
double random_double_scaled(double min, double max);
int    random_int(int min, int max);
double clamp(double x, double min, double max);
double deg2rad(double deg);

/* vec3 samplers with rand() draw order of vec3.c:56-107 */
vec3 vec3_random(void);
vec3 vec3_random_scaled(double min, double max);
vec3 vec3_random_in_unit_sphere(void);
vec3 vec3_random_unit_vector_in_unit_sphere(void);
vec3 random_in_unit_disk(void);
vec3 random_cosine_direction(void);
vec3 vec3c_refract(vec3 uv, vec3 n, double etai_over_etat);
vec3 vec3c_reflect(vec3 v, vec3 n);

point3 ray_at(ray *r, double t);
void   write_color(unsigned char *pixels, color *pixel_color, int samples_per_pixel, int index);
color  ray_color(ray *r, color *background, hittable_list *world, hittable_list *lights, int recursion_depth);

int  aabb_hit(aabb *box, ray *r, double tmin, double tmax);
aabb surrounding_box(aabb box0, aabb box1);
void onb_init_from_w(onb *o, vec3 *n);
vec3 onb_localv(onb *o, vec3 a);
void rotatePoint(vec3 original, double sin_theta, double cos_theta, rotation_axis axis, vec3 *output);

texture *texture_solid_color_init(color c);
texture *texture_solid_color_init_rgb(double r, double g, double b);
texture *texture_checker_init(texture *odd, texture *even);
texture *texture_checker_init_c(color odd, color even);
texture *texture_noise_init_scaled(double sc);
texture *texture_noise_init(void);
texture *texture_image_init(char *filename); /* binary PPM (P6) only in the oracle; JPEG via orc_texture_image_from_rgb */
texture *orc_texture_image_from_rgb(const unsigned char *rgb, int w, int h);
color    texture_value(texture *t, double u, double v, vec3 *p);
void     textures_free_all(void);
perlin  *perlin_init(void);
double   perlin_noise(perlin *p, point3 point);
double   perlin_turb(perlin *perl, point3 p);

material *material_lambertian_new(texture *t);
material *material_lambertian_new_from_color(color c);
material *material_metal_new(color a, double fuzz);
material *material_dielectric_new(double ir);
material *material_light_new(texture *t);
material *material_light_new_from_color(color c);
material *material_isotropic_new(texture *t);
material *material_isotropic_new_from_color(color c);
void      materials_free_all(void);
double    reflectance(double cosine, double ref_idx);
color     material_emitted(material *mat, struct hit_record *rec, double u, double v, point3 p);
int       material_scatter(material *mat, ray *r, struct hit_record *rec, scatter_record *srec);
/* stage-2/1 scatter (02 src/material.c:181-193): attenuation + scattered ray */
int       material_scatter_att(material *mat, ray *r, struct hit_record *rec, color *attenuation, ray *scattered);
double    material_scattering_pdf(material *mat, ray *r, struct hit_record *rec, ray *scattered);

hittable_list *hittable_list_new(int max_size);
hittable_list *hittable_list_new_no_gc(int max_size);
void  hittable_list_add(hittable_list *l, hittable *o);
int   hittable_list_index(hittable_list *l);
void *hittable_list_objects(hittable_list *l);
int   hittable_list_hit(hittable_list *l, ray *r, double tmin, double tmax, hit_record *rec);
int   hittable_list_bounding_box(hittable_list *l, double t0, double t1, aabb *output_box);
double hittable_list_pdf_value(hittable_list *l, point3 *orig, vec3 *v);
vec3  hittable_list_random(hittable_list *l, vec3 orig);
void  hittable_lists_free_all(void);

hittable *hittable_sphere_new(struct hittable_list *world, point3 center, double r, struct material *mat);
hittable *hittable_rect_new(struct hittable_list *world, double x0, double x1, double y0, double y1, double z0, double z1, double k, rect_axis axis, struct material *mat);
hittable *hittable_box_new(struct hittable_list *world, point3 min, point3 max, struct material *mat);
hittable *hittable_moving_sphere_new(struct hittable_list *world, point3 center0, point3 center1, double t0, double t1, double r, struct material *mat);
hittable *bvh_node_init(struct hittable_list *world, struct hittable_list *objects, double t0, double t1);
hittable *hittable_translate_init(struct hittable_list *world, hittable *obj, vec3 offset);
hittable *hittable_rotate_init(struct hittable_list *world, hittable *obj, double angle, rotation_axis axis);
hittable *hittable_constant_medium_init(struct hittable_list *world, hittable *b, double d, texture *a);
hittable *hittable_constant_medium_init_c(struct hittable_list *world, hittable *b, double d, color c);
hittable *hittable_flip_face_init(struct hittable_list *world, hittable *b);
void      hittable_free(hittable *o);
int       hittable_hit(hittable *o, ray *r, double tmin, double tmax, hit_record *rec);
int       hittable_bounding_box(hittable *o, double t0, double t1, aabb *output_box);
double    hittable_pdf_value(hittable *o, point3 orig, vec3 v);
vec3      hittable_random(hittable *o, point3 orig);
int       rect_hit(rect *rc, ray *r, double tmin, double tmax, hit_record *rec);
int       sphere_hit(sphere *s, ray *r, double tmin, double tmax, hit_record *rec);

pdf   *pdf_cosine_init(vec3 *w);
void   pdf_hittable_list_init_stack(pdf *wrapper, pdf_hittable_list *instance, hittable_list *obj, point3 origin);
void   pdf_mixture_init_stack(pdf *wrapper, pdf_mixture *instance, pdf *a, pdf *b, double ratio);
void   pdf_free(pdf *p);
double pdf_value(pdf *p, vec3 *direction);
vec3   pdf_generate(pdf *p);
double pdf_cosine_value(pdf_cosine *p, vec3 *dir);

camera *camera_new(point3 lookfrom, point3 lookat, vec3 vup, double vfov, double aspect_ratio, double aperture, double focus_dist, double time0, double time1);
void    camera_free(camera *c);
ray     camera_get_ray(camera *c, double s, double t);

/* ============================ oracle-only API ============================= */
/* Which snapshot's material/emission/camera semantics are active (default 3). */
void orc_set_stage(int stage);
int  orc_get_stage(void);

/* RNG source for random_double(): thread-local.
 *   ORC_RNG_GLIBC  : rand()/(RAND_MAX+1.0), the reference's generator (utils.h:23)
 *   ORC_RNG_PHILOX : Philox4x32-10 stream keyed (seed; pixel, sample), 32-bit uniforms */
enum { ORC_RNG_GLIBC = 0, ORC_RNG_PHILOX = 1 };
void orc_rng_mode(int mode);
void orc_rng_seek(uint64_t seed, uint32_t pixel, uint32_t sample);
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Integrator variants. integrator: 0 = stage-3 mixture PDF, 1 = stage-2 attenuation, 2 = stage-1 sky.
 * ray_color_iter is the loop form the GPU runs; it must agree with the recursion. */
typedef struct {
    int integrator;           /* 0,1,2 */
    int max_depth;
    double mixture_light_weight; /* 0.1, ray.c:72 */
    color background;
    int scrub_nan;            /* treat a non-finite sample as 0 (reference: no scrubbing, F16) */
} orc_settings;
color orc_ray_color(ray *r, const orc_settings *st, hittable_list *world, hittable_list *lights, int depth);
color orc_ray_color_iter(ray *r, const orc_settings *st, hittable_list *world, hittable_list *lights, uint64_t *rays_out);

/* Whole-image render into a double RGB sum buffer (W*H*3, row j=0 at the bottom
 * like the reference's pixel index main.c:157), samples [s_begin, s_end).
 * threads<=0 -> all cores. In GLIBC mode threads must be 1 for reproducibility. */
typedef struct { uint64_t samples, rays; double seconds; } orc_render_stats;
int orc_render(hittable_list *world, hittable_list *lights, camera *cam, const orc_settings *st,
               int width, int height, int s_begin, int s_end, int rng_mode, uint64_t seed,
               int threads, int iterative, double *rgb_sum, orc_render_stats *stats);

/* The reference's own loop order (16 row chunks, rows descending inside a chunk,
 * main.c:132-160) with rand(): byte-identical 8-bit output to the reference run single-threaded. */
/* 01_BasicTracer's output edge: write_color(FILE*, ...) (01 main.c:30-35) and the serial loop that feeds it, rows top to
 * bottom, one rand() stream continued from wherever it stands (01 main.c:168-191) */
void orc_write_color_ppm(FILE *f, color *pixel_color, int samples_per_pixel);
int  orc_render_ppm01(hittable_list *world, camera *cam, const orc_settings *st, int W, int H, int spp, const char *path);
void orc_render_refloop_u8(hittable_list *world, hittable_list *lights, camera *cam, const orc_settings *st,
                           int width, int height, int spp, unsigned char *pixels);

/* Event counters for the algorithmic FLOP model: thread-summed, reset by orc_counters_reset. */
typedef struct {
    uint64_t rays, miss, emit, diffuse, specular_metal, specular_dielectric, isotropic;
    uint64_t rect_tests, rect_hits, sphere_tests, sphere_hits, msphere_tests, aabb_tests, medium_tests;
    uint64_t translate_calls, rotate_calls, light_pdf_rect, light_pdf_sphere, light_sample_rect, light_sample_sphere;
    uint64_t cosine_sample, onb_builds, camera_rays, tex_noise, tex_checker, tex_image;
    uint64_t translate_hits, rotate_hits, light_pdf_rect_hits, light_pdf_sphere_hits, samples;
} orc_counters;
void orc_counters_reset(void);
void orc_counters_get(orc_counters *out);

#ifdef __cplusplus
}
#endif
#endif
