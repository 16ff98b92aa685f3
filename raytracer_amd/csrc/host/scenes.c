/*
 * scenes.c -- the named benchmark scenes, written once against the reference's
 * scene-construction API (hittable_*_new, material_*_new, texture_*_init,
 * hittable_list_*).  The SAME file is compiled three ways:
 *   - against include/raytracer.h          -> the product host library (librtc.so)
 *   - against oracle/cpu_ref.h             -> the CPU oracle (test infrastructure)
 *   - inside oracle/ref_shim/ref0{2,3}_tu.c against the reference's own headers
 *     where they lie in /root/reference    -> oracle/_ref/libref0{2,3}.so
 * which is the drop-in property the boundary promises.
 *
 * Scene constants restate /root/reference/03_FullTracer/main.c:34-80 (Cornell,
 * simple), 02_AdvancedTracer/src/main.c:76-289 (Book-2 zoo) and
 * 01_BasicTracer/src/main.c:72-116 (Book-1 random spheres); see SURVEY.md
 * Appendix B.  Scenes that the reference builds with rand() draw from rand() in
 * the same order, so srand(1) + build reproduces the reference's geometry.
 */
#ifdef RT_PRODUCT_BUILD          /* librtc.so only: the library looks for its image asset next to itself (dladdr) */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#endif
#ifdef RT_API_HEADER
#include RT_API_HEADER
#endif
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef RT_SCENES_TYPES
#define RT_SCENES_TYPES
/* Camera/integrator description that travels with a scene (the reference hard-codes these in main()). */
typedef struct rt_scene_view {
    point3 lookfrom, lookat;
    vec3 vup;
    double vfov, aspect, aperture, focus_dist, time0, time1;
    color background;
    int integrator;   /* 0 = 03 mixture-PDF, 1 = 02 attenuation, 2 = 01 sky gradient */
    int max_depth;
} rt_scene_view;
enum { RT_SCENE_SIMPLE = 0, RT_SCENE_CORNELL = 1, RT_SCENE_FINAL = 2, RT_SCENE_BOOK1 = 3,
       RT_SCENE_CORNELL02 = 4, RT_SCENE_CORNELL_SMOKE = 5, RT_SCENE_TWO_SPHERES = 6,
       RT_SCENE_TWO_PERLIN = 7, RT_SCENE_SIMPLE_LIGHT = 8, RT_SCENE_BOOK2_RANDOM = 9,
       RT_SCENE_MILLION = 10, RT_SCENE_EARTH = 11, RT_SCENE_FINAL_EARTH = 12, RT_SCENE_BOX_FIELD = 13, RT_SCENE_BOX_FIELD_SMOKE = 14 };
#endif

static vec3 sv3(double x, double y, double z) { vec3 v; v.x = x; v.y = y; v.z = z; return v; }

static void view_cornell(rt_scene_view *v) {
    v->lookfrom = sv3(278, 278, -800); v->lookat = sv3(278, 278, 0); v->vup = sv3(0, 1, 0);
    v->vfov = 40; v->aspect = 1.0; v->aperture = 0.0; v->focus_dist = 10; v->time0 = 0.0; v->time1 = 1.0;
    v->background = sv3(0, 0, 0);
}
static void view_book(rt_scene_view *v, double vfov, double aperture) {
    v->lookfrom = sv3(13, 2, 3); v->lookat = sv3(0, 0, 0); v->vup = sv3(0, 1, 0);
    v->vfov = vfov; v->aspect = 16.0 / 9.0; v->aperture = aperture; v->focus_dist = 10; v->time0 = 0.0; v->time1 = 1.0;
    v->background = sv3(0.70, 0.80, 1.00);
}

#ifndef RT_NO_FLIP_FACE
/* 03 main.c:67-80 -- one flipped area light over a red diffuse sphere */
static hittable_list *scene_simple(hittable_list *lights, rt_scene_view *v) {
    hittable_list *world = hittable_list_new(1024);
    material *red = material_lambertian_new_from_color(sv3(0.65, 0.05, 0.05));
    material *lamp = material_light_new_from_color(sv3(15, 15, 15));
    hittable *panel = hittable_rect_new(NULL, 213, 343, 0, 0, 227, 332, 554, XZ, lamp);
    panel = hittable_flip_face_init(world, panel);
    hittable_list_add(lights, panel);
    hittable_sphere_new(world, sv3(190, 190, 190), 90, red);
    view_cornell(v); v->integrator = 0; v->max_depth = 10;
    return world;
}
/* 03 main.c:34-63 -- Cornell box, aluminium block, glass ball; lights = {flipped panel, glass ball} */
static hittable_list *scene_cornell(hittable_list *lights, rt_scene_view *v) {
    hittable_list *world = hittable_list_new(1024);
    material *red = material_lambertian_new_from_color(sv3(0.65, 0.05, 0.05));
    material *white = material_lambertian_new_from_color(sv3(0.73, 0.73, 0.73));
    material *green = material_lambertian_new_from_color(sv3(0.12, 0.45, 0.15));
    material *lamp = material_light_new_from_color(sv3(15, 15, 15));
    hittable_rect_new(world, 0, 0, 0, 555, 0, 555, 555, YZ, green);
    hittable_rect_new(world, 0, 0, 0, 555, 0, 555, 0, YZ, red);
    hittable *panel = hittable_rect_new(NULL, 213, 343, 0, 0, 227, 332, 554, XZ, lamp);
    panel = hittable_flip_face_init(world, panel);
    hittable_list_add(lights, panel);
    hittable_rect_new(world, 0, 555, 0, 0, 0, 555, 0, XZ, white);
    hittable_rect_new(world, 0, 555, 0, 0, 0, 555, 555, XZ, white);
    hittable_rect_new(world, 0, 555, 0, 555, 0, 0, 555, XY, white);
    material *aluminium = material_metal_new(sv3(0.8, 0.85, 0.88), 0.0);
    hittable *block = hittable_box_new(NULL, sv3(0, 0, 0), sv3(165, 330, 165), aluminium);
    block = hittable_rotate_init(NULL, block, 15, Y);
    block = hittable_translate_init(world, block, sv3(265, 0, 295));
    material *glass = material_dielectric_new(1.5);
    hittable *ball = hittable_sphere_new(world, sv3(190, 90, 190), 90, glass);
    hittable_list_add(lights, ball);
    view_cornell(v); v->integrator = 0; v->max_depth = 10;
    return world;
}
#endif

/* Image asset of the earth scenes.  The reference names "images/earthmap.jpg" relative to its working directory (02 main.c:159).
 * Here the file is a datum of the library: rt_scene_set_image_path() names it explicitly; otherwise $RT_EARTHMAP, then
 * <directory of librtc.so>/../share/raytracer_amd/earthmap.jpg (`make install` puts it there), then <directory of librtc.so>/assets/
 * earthmap.jpg (this source tree: raytracer_amd/assets/). */
static char g_rt_image_path[1024] = "";
void rt_scene_set_image_path(const char *path) { if (path) { strncpy(g_rt_image_path, path, sizeof g_rt_image_path - 1); g_rt_image_path[sizeof g_rt_image_path - 1] = 0; } }
static char *rt_image_path(void) {
    if (g_rt_image_path[0]) return g_rt_image_path;
    static char found[1100];
    const char *env = getenv("RT_EARTHMAP");
    if (env && env[0]) { snprintf(found, sizeof found, "%s", env); return found; }
#ifdef RT_PRODUCT_BUILD
    Dl_info info;
    if (dladdr((const void *)&rt_scene_set_image_path, &info) && info.dli_fname) {
        char dir[1024]; snprintf(dir, sizeof dir, "%s", info.dli_fname);
        char *slash = strrchr(dir, '/');
        if (slash) *slash = 0; else snprintf(dir, sizeof dir, ".");
        const char *tails[2] = {"/../share/raytracer_amd/earthmap.jpg", "/assets/earthmap.jpg"};
        for (int k = 0; k < 2; k++) {
            snprintf(found, sizeof found, "%s%s", dir, tails[k]);
            FILE *f = fopen(found, "rb");
            if (f) { fclose(f); return found; }
        }
    }
#endif
    snprintf(found, sizeof found, "earthmap.jpg");            /* texture_image_init reports the failure like the reference (texture.c:67) */
    return found;
}

/* 02 main.c:230-289 -- Book-2 final scene.  rand() order: 400 box heights, BVH nodes,
 * perlin tables, 3000 sphere coordinates, BVH nodes. */
static hittable_list *scene_final(hittable_list *lights, rt_scene_view *v, int with_earth) {
    hittable_list *ground_boxes = hittable_list_new_no_gc(1024);
    material *ground = material_lambertian_new_from_color(sv3(0.48, 0.83, 0.53));
    for (int i = 0; i < 20; i++)
        for (int j = 0; j < 20; j++) {
            double w = 100.0, x0 = -1000.0 + i * w, z0 = -1000.0 + j * w, y0 = 0.0;
            double x1 = x0 + w, y1 = random_double_scaled(1, 101), z1 = z0 + w;
            hittable_box_new(ground_boxes, sv3(x0, y0, z0), sv3(x1, y1, z1), ground);
        }
    hittable_list *objects = hittable_list_new(1024);
    bvh_node_init(objects, ground_boxes, 0, 1);
    material *lamp = material_light_new_from_color(sv3(7, 7, 7));
    hittable *panel = hittable_rect_new(objects, 123, 423, 0, 0, 147, 412, 554, XZ, lamp);
    hittable_list_add(lights, panel);
    point3 c1 = sv3(400, 400, 200), c2 = sv3(430, 400, 200);
    hittable_moving_sphere_new(objects, c1, c2, 0, 1, 50, material_lambertian_new_from_color(sv3(0.7, 0.3, 0.1)));
    hittable_sphere_new(objects, sv3(260, 150, 45), 50, material_dielectric_new(1.5));
    hittable_sphere_new(objects, sv3(0, 150, 145), 50, material_metal_new(sv3(0.8, 0.8, 0.8), 1.0));
    hittable *boundary = hittable_sphere_new(NULL, sv3(360, 150, 145), 70, material_dielectric_new(1.5));
    hittable_constant_medium_init_c(objects, boundary, 0.025, sv3(0.2, 0.4, 0.9));
    boundary = hittable_sphere_new(NULL, sv3(0, 0, 0), 5000, material_dielectric_new(1.5));
    hittable_constant_medium_init_c(objects, boundary, 0.0001, sv3(1, 1, 1));
    texture *marble = texture_noise_init_scaled(5);
    hittable_sphere_new(objects, sv3(220, 280, 300), 80, material_lambertian_new(marble));
    if (with_earth) { /* BASELINE config 3 names an earth-map globe; the reference's final_scene has none: book position */
        texture *earth = texture_image_init(rt_image_path());
        if (!earth) return NULL;
        hittable_sphere_new(objects, sv3(400, 200, 400), 100, material_lambertian_new(earth));
    }
    hittable_list *cloud = hittable_list_new_no_gc(1024);
    material *white = material_lambertian_new_from_color(sv3(0.73, 0.73, 0.73));
    for (int j = 0; j < 1000; j++) {
        double a = random_double_scaled(0, 165), b = random_double_scaled(0, 165), c = random_double_scaled(0, 165);
        hittable_sphere_new(cloud, sv3(a, b, c), 10, white);
    }
    hittable *bvh = bvh_node_init(NULL, cloud, 0, 1);
    hittable *rotated = hittable_rotate_init(NULL, bvh, 15, Y);
    hittable_translate_init(objects, rotated, sv3(-100, 270, 395));
    view_cornell(v); v->lookfrom = sv3(478, 278, -600);
    v->integrator = 1; v->max_depth = 8;
    return objects;
}

/* 01 main.c:72-116 -- Book-1 random spheres (static spheres, sky background). */
static hittable_list *scene_book1(rt_scene_view *v) {
    hittable_list *world = hittable_list_new(1024);
    hittable_sphere_new(world, sv3(0, -1000, 0), 1000, material_lambertian_new_from_color(sv3(0.5, 0.5, 0.5)));
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            double choose = random_double();
            double cx = a + 0.9 * random_double(), cz = b + 0.9 * random_double();
            point3 center = sv3(cx, 0.2, cz);
            double dx = center.x - 4, dy = center.y - 0.2, dz = center.z - 0;
            if (sqrt((dx * dx) + (dy * dy) + (dz * dz)) > 0.9) {
                if (choose < 0.8) {
                    double a0 = random_double(), a1 = random_double(), a2 = random_double();
                    double b0 = random_double(), b1 = random_double(), b2 = random_double();
                    hittable_sphere_new(world, center, 0.2, material_lambertian_new_from_color(sv3(a0 * b0, a1 * b1, a2 * b2)));
                } else if (choose < 0.95) {
                    double a0 = random_double_scaled(0.5, 1), a1 = random_double_scaled(0.5, 1), a2 = random_double_scaled(0.5, 1);
                    double fuzz = random_double_scaled(0, 0.5);
                    hittable_sphere_new(world, center, 0.2, material_metal_new(sv3(a0, a1, a2), fuzz));
                } else {
                    hittable_sphere_new(world, center, 0.2, material_dielectric_new(1.5));
                }
            }
        }
    hittable_sphere_new(world, sv3(0, 1, 0), 1.0, material_dielectric_new(1.5));
    hittable_sphere_new(world, sv3(-4, 1, 0), 1.0, material_lambertian_new_from_color(sv3(0.4, 0.2, 0.1)));
    hittable_sphere_new(world, sv3(4, 1, 0), 1.0, material_metal_new(sv3(0.7, 0.6, 0.5), 0.0));
    view_book(v, 30, 0.1); v->integrator = 2; v->max_depth = 8;
    return world;
}

/* 02 main.c:180-201 -- Book-2 Cornell box with two white blocks (first block rotated with axis X, F10). */
static hittable_list *scene_cornell02(int smoke, rt_scene_view *v) {
    hittable_list *world = hittable_list_new(1024);
    material *red = material_lambertian_new_from_color(sv3(0.65, 0.05, 0.05));
    material *white = material_lambertian_new_from_color(sv3(0.73, 0.73, 0.73));
    material *green = material_lambertian_new_from_color(sv3(0.12, 0.45, 0.15));
    material *lamp = material_light_new_from_color(sv3(15, 15, 15));
    hittable_rect_new(world, 0, 0, 0, 555, 0, 555, 555, YZ, green);
    hittable_rect_new(world, 0, 0, 0, 555, 0, 555, 0, YZ, red);
    hittable_rect_new(world, 213, 343, 0, 0, 227, 332, 554, XZ, lamp);
    hittable_rect_new(world, 0, 555, 0, 0, 0, 555, 0, XZ, white);
    hittable_rect_new(world, 0, 555, 0, 0, 0, 555, 555, XZ, white);
    hittable_rect_new(world, 0, 555, 0, 555, 0, 0, 555, XY, white);
    hittable *tall = hittable_box_new(NULL, sv3(0, 0, 0), sv3(165, 330, 165), white);
    hittable *small = hittable_box_new(NULL, sv3(0, 0, 0), sv3(165, 165, 165), white);
    if (!smoke) { /* :194-199 */
        tall = hittable_rotate_init(NULL, tall, 15, X);
        hittable_translate_init(world, tall, sv3(265, 0, 295));
        small = hittable_rotate_init(NULL, small, -18, Y);
        hittable_translate_init(world, small, sv3(130, 0, 65));
    } else { /* :217-224 */
        tall = hittable_rotate_init(NULL, tall, 15, Y);
        tall = hittable_translate_init(NULL, tall, sv3(265, 0, 295));
        hittable_constant_medium_init_c(world, tall, 0.01, sv3(0, 0, 0));
        small = hittable_rotate_init(NULL, small, -18, Y);
        small = hittable_translate_init(NULL, small, sv3(130, 0, 65));
        hittable_constant_medium_init_c(world, small, 0.01, sv3(1, 1, 1));
    }
    view_cornell(v); v->integrator = 1; v->max_depth = 8;
    return world;
}

/* Not a scene of the reference: the 02 Cornell room (main.c:186-199) filled with 18 boxes under rotate_x / rotate_y /
 * rotate_z + translate and 8 spheres.  114 leaves, so it is flattened into a BVH whose rect leaves are ROTATED -- the
 * case none of the reference's own scenes puts into a BVH.  Deterministic (no rand()).  smoke: plus two constant media
 * (02 main.c:217-224 style), one bounded by a rotated box, so that a BVH scene carries a general medium. */
static hittable_list *scene_box_field(rt_scene_view *v, int smoke) {
    hittable_list *world = hittable_list_new(1024);
    material *red = material_lambertian_new_from_color(sv3(0.65, 0.05, 0.05));
    material *white = material_lambertian_new_from_color(sv3(0.73, 0.73, 0.73));
    material *green = material_lambertian_new_from_color(sv3(0.12, 0.45, 0.15));
    material *lamp = material_light_new_from_color(sv3(15, 15, 15));
    material *steel = material_metal_new(sv3(0.8, 0.85, 0.88), 0.1);
    material *glass = material_dielectric_new(1.5);
    hittable_rect_new(world, 0, 0, 0, 555, 0, 555, 555, YZ, green);
    hittable_rect_new(world, 0, 0, 0, 555, 0, 555, 0, YZ, red);
    hittable_rect_new(world, 163, 393, 0, 0, 177, 382, 554, XZ, lamp);
    hittable_rect_new(world, 0, 555, 0, 0, 0, 555, 0, XZ, white);
    hittable_rect_new(world, 0, 555, 0, 0, 0, 555, 555, XZ, white);
    hittable_rect_new(world, 0, 555, 0, 555, 0, 0, 555, XY, white);
    for (int i = 0; i < 18; i++) {
        const int gx = i % 6, gz = i / 6;
        const double w = 40 + 7 * ((i * 5) % 4), hgt = 50 + 23 * ((i * 7) % 5), dep = 40 + 9 * ((i * 3) % 4);
        hittable *b = hittable_box_new(NULL, sv3(0, 0, 0), sv3(w, hgt, dep), i % 5 == 0 ? steel : white);
        b = hittable_rotate_init(NULL, b, -40.0 + 13.0 * ((i * 11) % 7), i % 3 == 0 ? Y : (i % 3 == 1 ? X : Z));
        hittable_translate_init(world, b, sv3(45 + 85 * gx, 20 + 30 * (i % 2), 120 + 150 * gz));
    }
    for (int i = 0; i < 8; i++)
        hittable_sphere_new(world, sv3(70 + 60 * i, 330 + 25 * (i % 3), 200 + 35 * ((i * 3) % 5)), 22 + 3 * (i % 4), i % 2 ? glass : steel);
    if (smoke) {   /* a box-bounded medium (general boundary: two list queries per visit) and a sphere-bounded one, inside the BVH */
        hittable *b = hittable_box_new(NULL, sv3(0, 0, 0), sv3(150, 200, 150), white);
        b = hittable_rotate_init(NULL, b, 20, Y);
        b = hittable_translate_init(NULL, b, sv3(200, 150, 250));
        hittable_constant_medium_init_c(world, b, 0.008, sv3(0.1, 0.1, 0.1));
        hittable *ball = hittable_sphere_new(NULL, sv3(400, 300, 150), 90, white);
        hittable_constant_medium_init_c(world, ball, 0.01, sv3(0.9, 0.9, 1.0));
    }
    view_cornell(v); v->integrator = 1; v->max_depth = 8;
    return world;
}

/* 02 main.c:132-176 -- checker / perlin / simple-light scenes */
static hittable_list *scene_two_spheres(rt_scene_view *v) {
    hittable_list *world = hittable_list_new(1024);
    texture *tex = texture_checker_init(texture_solid_color_init_rgb(0.5, 0.5, 0.5), texture_solid_color_init_rgb(0.2, 0.2, 0.2));
    material *m = material_lambertian_new(tex);
    hittable_sphere_new(world, sv3(0, -10, 0), 10, m);
    hittable_sphere_new(world, sv3(0, 10, 0), 10, m);
    view_book(v, 20, 0.0); v->integrator = 1; v->max_depth = 8;
    return world;
}
static hittable_list *scene_perlin(int with_light, rt_scene_view *v) {
    hittable_list *world = hittable_list_new(1024);
    material *m = material_lambertian_new(texture_noise_init_scaled(4));
    hittable_sphere_new(world, sv3(0, -1000, 0), 1000, m);
    hittable_sphere_new(world, sv3(0, 2, 0), 2, m);
    view_book(v, 20, 0.0); v->integrator = 1; v->max_depth = 8;
    if (with_light) {
        hittable_rect_new(world, 3, 5, 1, 3, 0, 0, -2, XY, material_light_new_from_color(sv3(4, 4, 4)));
        v->background = sv3(0, 0, 0); v->lookfrom = sv3(26, 3, 6); v->lookat = sv3(0, 2, 0);
    }
    return world;
}
/* 02 main.c:157-163 with the camera of its switch case 4 (:344-350) -- one globe with the earth map */
static hittable_list *scene_earth(rt_scene_view *v) {
    hittable_list *world = hittable_list_new(1024);
    texture *earth = texture_image_init(rt_image_path());
    if (!earth) return NULL;
    hittable_sphere_new(world, sv3(0, 0, 0), 2, material_lambertian_new(earth));
    view_book(v, 20, 0.0); v->integrator = 1; v->max_depth = 8;
    return world;
}
/* 02 main.c:76-128 -- Book-2 random scene: checker ground, moving diffuse spheres */
static hittable_list *scene_book2_random(rt_scene_view *v) {
    hittable_list *world = hittable_list_new(1024);
    texture *checker_tex = texture_checker_init(texture_solid_color_init_rgb(0.5, 0.5, 0.5), texture_solid_color_init_rgb(0.2, 0.2, 0.2));
    texture *ball_tex = texture_solid_color_init_rgb(0.4, 0.2, 0.1);
    hittable_sphere_new(world, sv3(0, -1000, 0), 1000, material_lambertian_new(checker_tex));
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            double choose = random_double();
            double cx = a + 0.9 * random_double(), cz = b + 0.9 * random_double();
            point3 center = sv3(cx, 0.2, cz);
            double dx = center.x - 4, dy = center.y - 0.2, dz = center.z - 0;
            if (sqrt((dx * dx) + (dy * dy) + (dz * dz)) > 0.9) {
                if (choose < 0.8) {
                    double a0 = random_double(), a1 = random_double(), a2 = random_double();
                    double b0 = random_double(), b1 = random_double(), b2 = random_double();
                    material *m = material_lambertian_new_from_color(sv3(a0 * b0, a1 * b1, a2 * b2));
                    point3 center2 = sv3(center.x + 0, center.y + random_double_scaled(0, 0.5), center.z + 0);
                    hittable_moving_sphere_new(world, center, center2, 0.0, 1.0, 0.2, m);
                } else if (choose < 0.95) {
                    double a0 = random_double_scaled(0.5, 1), a1 = random_double_scaled(0.5, 1), a2 = random_double_scaled(0.5, 1);
                    double fuzz = random_double_scaled(0, 0.5);
                    hittable_sphere_new(world, center, 0.2, material_metal_new(sv3(a0, a1, a2), fuzz));
                } else {
                    hittable_sphere_new(world, center, 0.2, material_dielectric_new(1.5));
                }
            }
        }
    hittable_sphere_new(world, sv3(0, 1, 0), 1.0, material_dielectric_new(1.5));
    hittable_sphere_new(world, sv3(-4, 1, 0), 1.0, material_lambertian_new(ball_tex));
    hittable_sphere_new(world, sv3(4, 1, 0), 1.0, material_metal_new(sv3(0.7, 0.6, 0.5), 0.0));
    view_book(v, 40, 0.1); v->integrator = 1; v->max_depth = 8;
    return world;
}

/* BASELINE.json config 5 -- synthetic n_side x n_side grid of r=0.2 spheres with the Book-1 material mix
 * (80% diffuse / 15% metal / 5% glass), inside a BVH.  Uses its own 64-bit LCG so the scene does not
 * depend on glibc.  SURVEY.md section 8d, config 5. */
static unsigned long long mil_state;
static double mil_rand(void) { mil_state = mil_state * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(mil_state >> 11) * (1.0 / 9007199254740992.0); }
static hittable_list *scene_million(int n_side, rt_scene_view *v) {
    hittable_list *world = hittable_list_new(16);
    hittable_list *grid = hittable_list_new_no_gc(n_side * n_side + 8);
    mil_state = 0x5EEDULL;
    /* a few hundred shared materials keep the material table small without changing the mix */
    enum { NPAL = 256 };
    material *pal_diffuse[NPAL], *pal_metal[NPAL];
    for (int i = 0; i < NPAL; i++) {
        pal_diffuse[i] = material_lambertian_new_from_color(sv3(mil_rand() * mil_rand(), mil_rand() * mil_rand(), mil_rand() * mil_rand()));
        pal_metal[i] = material_metal_new(sv3(0.5 + 0.5 * mil_rand(), 0.5 + 0.5 * mil_rand(), 0.5 + 0.5 * mil_rand()), 0.5 * mil_rand());
    }
    material *glass = material_dielectric_new(1.5);
    int half = n_side / 2;
    for (int a = -half; a < n_side - half; a++)
        for (int b = -half; b < n_side - half; b++) {
            double choose = mil_rand();
            point3 center = sv3(a + 0.9 * mil_rand(), 0.2, b + 0.9 * mil_rand());
            int pick = (int)(mil_rand() * NPAL) & (NPAL - 1);
            material *m = choose < 0.8 ? pal_diffuse[pick] : (choose < 0.95 ? pal_metal[pick] : glass);
            hittable_sphere_new(grid, center, 0.2, m);
        }
    hittable_sphere_new(grid, sv3(0, 1, 0), 1.0, glass);
    hittable_sphere_new(grid, sv3(-4, 1, 0), 1.0, material_lambertian_new_from_color(sv3(0.4, 0.2, 0.1)));
    hittable_sphere_new(grid, sv3(4, 1, 0), 1.0, material_metal_new(sv3(0.7, 0.6, 0.5), 0.0));
    hittable_sphere_new(world, sv3(0, -1000, 0), 1000, material_lambertian_new_from_color(sv3(0.5, 0.5, 0.5)));
    bvh_node_init(world, grid, 0, 1);
    view_book(v, 30, 0.1); v->integrator = 2; v->max_depth = 8;
    return world;
}

/* One entry point for every backend.  `arg` is scene-specific (RT_SCENE_MILLION: spheres per side). */
hittable_list *rt_scene_build(int id, int arg, hittable_list *lights, rt_scene_view *view) {
    switch (id) {
#ifndef RT_NO_FLIP_FACE
    case RT_SCENE_SIMPLE: return scene_simple(lights, view);
    case RT_SCENE_CORNELL: return scene_cornell(lights, view);
#endif
    case RT_SCENE_FINAL: return scene_final(lights, view, 0);
    case RT_SCENE_FINAL_EARTH: return scene_final(lights, view, 1);
    case RT_SCENE_EARTH: return scene_earth(view);
    case RT_SCENE_BOOK1: return scene_book1(view);
    case RT_SCENE_CORNELL02: return scene_cornell02(0, view);
    case RT_SCENE_CORNELL_SMOKE: return scene_cornell02(1, view);
    case RT_SCENE_TWO_SPHERES: return scene_two_spheres(view);
    case RT_SCENE_TWO_PERLIN: return scene_perlin(0, view);
    case RT_SCENE_SIMPLE_LIGHT: return scene_perlin(1, view);
    case RT_SCENE_BOOK2_RANDOM: return scene_book2_random(view);
    case RT_SCENE_MILLION: return scene_million(arg > 0 ? arg : 1000, view);
    case RT_SCENE_BOX_FIELD: return scene_box_field(view, 0);
    case RT_SCENE_BOX_FIELD_SMOKE: return scene_box_field(view, 1);
    default: return NULL;
    }
}
