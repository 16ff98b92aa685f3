/*
 * scene_internal.h -- host-side object graph behind include/raytracer.h (product code).
 *
 * The graph only has to remember WHAT the scene code asked for; it is never traversed for rendering.
 * rtc_scene_build (flatten.cpp) walks it once and emits flat tables for the GPU.
 */
#ifndef RAYTRACER_AMD_SCENE_INTERNAL_H
#define RAYTRACER_AMD_SCENE_INTERNAL_H

#include <stddef.h>

#include "raytracer.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { point3 lo, hi; } rt_box;

typedef enum { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_NOISE = 2, RT_TEX_IMAGE = 3 } rt_tex_kind;
struct texture {
    rt_tex_kind kind;
    color value;               /* solid */
    texture *odd, *even;       /* checker */
    double scale;              /* noise */
    double *ranvec;            /* noise: 256 x 3 */
    int *perm;                 /* noise: 3 x 256 (x, y, z) */
    unsigned char *pixels;     /* image: RGB8 */
    int width, height;
    int id;                    /* registry slot */
};

typedef enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2, RT_MAT_LIGHT = 3, RT_MAT_ISOTROPIC = 4 } rt_mat_kind;
struct material {
    rt_mat_kind kind;
    texture *tex;              /* lambertian albedo / light emit / isotropic albedo */
    color albedo;              /* metal */
    double param;              /* metal fuzz / dielectric ir */
    int id;
};

typedef enum { RT_H_SPHERE, RT_H_MOVING_SPHERE, RT_H_RECT, RT_H_BOX, RT_H_BVH, RT_H_TRANSLATE, RT_H_ROTATE,
               RT_H_MEDIUM, RT_H_FLIP } rt_hit_kind;
struct hittable {
    rt_hit_kind kind;
    material *mat;             /* leaves; medium: phase function */
    /* sphere: c0,r | moving sphere: c0,c1,t0,t1,r | rect: lo/hi (unused pair = 0), k, axis | box: lo,hi */
    point3 c0, c1;
    double r, t0, t1, k;
    point3 lo, hi;
    int axis;                  /* rect_axis or rotation_axis */
    hittable *child;           /* translate / rotate / flip / medium boundary */
    hittable_list *members;    /* box sides (owned), bvh source list (not owned) */
    int *bvh_order;            /* bvh: member indices in the reference's leaf visit order */
    int bvh_count;
    vec3 offset;               /* translate */
    double sin_theta, cos_theta; /* rotate */
    int hasbox; rt_box bbox;   /* rotate: precomputed like hittable.c:255-284 */
    double neg_inv_density;    /* medium */
};

struct hittable_list { hittable **objs; int index, size, gc; };

struct camera {
    point3 origin, lower_left_corner;
    vec3 horizontal, vertical, u, v, w;
    double lens_radius, time0, time1;
};

int rt_bounding_box(const hittable *o, double t0, double t1, rt_box *out);
/* jpeg_decode.c: baseline JPEG -> malloc'ed RGB8, rows top to bottom; 0 on malformed / unsupported input */
/* changes whenever anything is constructed, added to a list or freed through the API (scene_api.c) */
unsigned long long rt_api_generation(void);
int rt_jpeg_decode(const unsigned char *data, size_t n, int *width, int *height, unsigned char **rgb_out);
int rt_png_decode(const unsigned char *data, size_t n, int *width, int *height, unsigned char **rgb_out);   /* image_decode.c */
int rt_bmp_decode(const unsigned char *data, size_t n, int *width, int *height, unsigned char **rgb_out);

#ifdef __cplusplus
}
#endif
#endif
