/*
 * include/raytracer.h -- host-side scene-construction API of raytracer_amd.
 *
 * Source-compatible with the construction surface of elkiwy/raytraCer's
 * 03_FullTracer (and 02_AdvancedTracer): the scene code of the reference's
 * main.c (setupScene, simpleScene, final_scene, ...) compiles unchanged against
 * this header.  Every declaration cites the reference declaration it replaces
 * (paths under /root/reference/).  Unlike the reference, objects are opaque:
 * the CPU evaluation half of the reference's API (hittable_hit, material_scatter,
 * pdf_*, ray_color) is NOT provided on the host -- that work runs on the GPU
 * behind include/rtc.h, and there is no CPU fallback.
 *
 * Ownership (same as the reference): constructors return heap
 * objects; `world == NULL` means "do not append to any list"; a list from
 * hittable_list_new owns its members, one from hittable_list_new_no_gc does not;
 * wrappers own their child; materials and textures belong to global registries
 * released by materials_free_all / textures_free_all.  The registries here are
 * growable (the reference's are fixed 1024-slot arrays, material.h:50,
 * texture.h:66, hittable_list.h:34) and hittable_list_add grows the list
 * (the reference has no bounds check, hittable_list.c:97-101).
 */
#ifndef RAYTRACER_AMD_RAYTRACER_H
#define RAYTRACER_AMD_RAYTRACER_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef PI
#define PI 3.1415926535897932385 /* 03_FullTracer/utils.h:17 */
#endif

/* 03_FullTracer/vec3.h:17,83-84 */
typedef struct { double x, y, z; } vec3;
typedef vec3 point3;
typedef vec3 color;
/* 03_FullTracer/ray.h:21 */
typedef struct { point3 orig; vec3 dir; double time; } ray;

/* opaque handles: 03_FullTracer/hittable.h:55-58, hittable_list.h:20-25, material.h:28, texture.h:20, camera.h:17-25 */
typedef struct hittable hittable;
typedef struct hittable_list hittable_list;
typedef struct material material;
typedef struct texture texture;
typedef struct camera camera;

/* 03_FullTracer/hittable.h:69,103 */
typedef enum { XY, XZ, YZ } rect_axis;
typedef enum { X, Y, Z } rotation_axis; /* X rotates the (x,y) pair, Y (x,z), Z (y,z): reference quirk F10, hittable.c:37-48 */

/* ---- utils (03_FullTracer/utils.h:21-30): rand()-based, kept for scene code that draws random layouts */
double random_double(void);
double random_double_scaled(double min, double max);
int    random_int(int min, int max);
double deg2rad(double deg);

/* ---- textures (03_FullTracer/texture.h:56-68) */
texture *texture_solid_color_init(color c);
texture *texture_solid_color_init_rgb(double r, double g, double b);
texture *texture_checker_init(texture *odd, texture *even);
texture *texture_checker_init_c(color odd, color even);
texture *texture_noise_init_scaled(double sc); /* builds the perlin tables with rand(), perlin.c:67-75 */
texture *texture_noise_init(void);
texture *texture_image_init(char *filename);   /* JPEG (baseline, progressive), PNG, BMP, binary PGM / PPM; NULL + message on failure like texture.c:67 */
texture *texture_image_init_rgb(const unsigned char *rgb, int width, int height); /* extension: decoded pixels */
void     texture_free(texture *t);              /* texture.h:65; also clears the registry slot (the reference leaves it dangling) */
void     textures_free_all(void);

/* ---- materials (03_FullTracer/material.h:39-52) */
material *material_lambertian_new(texture *t);
material *material_lambertian_new_from_color(color c);
material *material_metal_new(color a, double fuzz);
material *material_dielectric_new(double ir);
material *material_light_new(texture *t);
material *material_light_new_from_color(color c);
material *material_isotropic_new(texture *t);
material *material_isotropic_new_from_color(color c);
void      material_free(material *m);           /* material.h:47; also clears the registry slot */
void      materials_free_all(void);

/* ---- lists (03_FullTracer/hittable_list.h:29-41) */
hittable_list *hittable_list_new(int max_size);
hittable_list *hittable_list_new_no_gc(int max_size);
void  hittable_list_add(hittable_list *l, hittable *o);
int   hittable_list_index(hittable_list *l);
void *hittable_list_objects(hittable_list *l);
void  hittable_list_free(hittable_list *l);
void  hittable_lists_free_all(void);

/* ---- hittables (03_FullTracer/hittable.h:117-129) */
hittable *hittable_sphere_new(hittable_list *world, point3 center, double r, material *mat);
hittable *hittable_rect_new(hittable_list *world, double x0, double x1, double y0, double y1, double z0, double z1,
                            double k, rect_axis axis, material *mat);
hittable *hittable_box_new(hittable_list *world, point3 min, point3 max, material *mat);
hittable *hittable_moving_sphere_new(hittable_list *world, point3 center0, point3 center1, double t0, double t1,
                                     double r, material *mat);
/* Records the member list; the device BVH is built at rtc_scene_build time.  Consumes one rand() per node the
 * reference's constructor would create (hittable.c:172) so that rand()-driven scene code stays in step. */
hittable *bvh_node_init(hittable_list *world, hittable_list *objects, double t0, double t1);
hittable *hittable_translate_init(hittable_list *world, hittable *obj, vec3 offset);
hittable *hittable_rotate_init(hittable_list *world, hittable *obj, double angle, rotation_axis axis);
hittable *hittable_constant_medium_init(hittable_list *world, hittable *b, double d, texture *a);
hittable *hittable_constant_medium_init_c(hittable_list *world, hittable *b, double d, color c);
hittable *hittable_flip_face_init(hittable_list *world, hittable *b);
void      hittable_free(hittable *o);

/* ---- camera (03_FullTracer/camera.h:29-32).  Fields are readable through camera_fields(). */
camera *camera_new(point3 lookfrom, point3 lookat, vec3 vup, double vfov, double aspect_ratio, double aperture,
                   double focus_dist, double time0, double time1);
void    camera_free(camera *c);
/* camera.h:35 / camera.c:46-53: one primary ray on the host, lens sample and ray time drawn from rand() like the
 * reference (F13).  The renderer generates its primary rays on the device; this is for callers that want single rays. */
ray     camera_get_ray(camera *c, double s, double t);
/* origin, lower_left_corner, horizontal, vertical, u, v, w (7 x 3), lens_radius, time0, time1 = 24 doubles */
void    camera_fields(const camera *c, double out[24]);

/* ---- the integrator's host-visible ends (03_FullTracer/ray.h:36-39)
 * write_color: ray.c:35-47 on the host, bytes identical to the reference's (NaN -> 0 instead of undefined behaviour).
 * ray_color:   ray.c:50-84.  ONE sample of the estimator for ONE ray, traced on the GPU (rtc_ray_colors, include/rtc.h):
 *              the scene tables of (world, lights, background) are built on the first call and kept until another
 *              world / lights / list length / background is passed; rand() is replaced by a Philox stream that advances
 *              with every call.  Returns (0,0,0) and prints rtc_last_error() when no GPU is usable: there is no CPU path. */
void  write_color(unsigned char *pixels, color *pixel_color, int samples_per_pixel, int index);
color ray_color(ray *r, color *background, hittable_list *world, hittable_list *lights, int recursion_depth);

/* ---- named scenes (raytracer_amd/csrc/host/scenes.c) */
typedef struct rt_scene_view {
    point3 lookfrom, lookat;
    vec3 vup;
    double vfov, aspect, aperture, focus_dist, time0, time1;
    color background;
    int integrator; /* 0 = 03 mixture-PDF (ray.c:50), 1 = 02 attenuation (02 main.c:44), 2 = 01 sky (01 main.c:39) */
    int max_depth;
} rt_scene_view;
enum { RT_SCENE_SIMPLE = 0, RT_SCENE_CORNELL = 1, RT_SCENE_FINAL = 2, RT_SCENE_BOOK1 = 3,
       RT_SCENE_CORNELL02 = 4, RT_SCENE_CORNELL_SMOKE = 5, RT_SCENE_TWO_SPHERES = 6,
       RT_SCENE_TWO_PERLIN = 7, RT_SCENE_SIMPLE_LIGHT = 8, RT_SCENE_BOOK2_RANDOM = 9,
       RT_SCENE_MILLION = 10, RT_SCENE_EARTH = 11, RT_SCENE_FINAL_EARTH = 12,
       RT_SCENE_BOX_FIELD = 13, RT_SCENE_BOX_FIELD_SMOKE = 14 /* not reference scenes: rotated boxes (and media) inside a BVH (host/scenes.c) */ };
#define RT_SCENES_TYPES 1
hittable_list *rt_scene_build(int id, int arg, hittable_list *lights, rt_scene_view *view);
/* image file of RT_SCENE_EARTH / RT_SCENE_FINAL_EARTH (the reference hard-codes "images/earthmap.jpg", 02 main.c:159);
 * JPEG or binary PPM; default "tests/golden/earthmap.jpg" (the reference's own 1024x512 asset) relative to the working directory */
void rt_scene_set_image_path(const char *path);

/* ---- image output: 8-bit RGB rows top-to-bottom -> PNG.  flip_vertically=1 reproduces
 * stbi_flip_vertically_on_write(1) + stbi_write_png(path,w,h,3,pixels,w*3) of 03_FullTracer/main.c:170-171
 * (same decoded pixels; the deflate stream is zlib's, not stb's). Returns 1 on success like stbi_write_png. */
int rt_write_png(const char *path, int w, int h, const unsigned char *rgb, int flip_vertically);
/* 01_BasicTracer's ASCII PPM (P3), rows written top to bottom (01 main.c:168,172-191) */
int rt_write_ppm_p3(const char *path, int w, int h, const unsigned char *rgb, int flip_vertically);

#ifdef __cplusplus
}
#endif
#endif
