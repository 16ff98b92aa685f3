/*
 * oracle/harness.c -- flat, ctypes-friendly probes (TEST INFRASTRUCTURE).
 *
 * One source, three builds (see oracle/Makefile):
 *   HX_ORACLE        : against oracle/cpu_ref.h              -> oracle/_build/liboracle.so
 *   HX_STAGE == 3    : textually after the reference's 03_FullTracer/main.c (ref_shim/ref03_tu.c)
 *   HX_STAGE == 2    : textually after the reference's 02_AdvancedTracer/src/main.c (ref_shim/ref02_tu.c)
 * The reference builds call the reference's OWN functions; nothing here re-implements
 * hit/scatter/pdf maths.  The only logic of its own is the leaf walker that assigns
 * primitive ids (the reference's hit_record carries none): it re-tests single
 * leaves with the reference's hittable_hit and keeps the last leaf whose t equals the
 * authoritative t from hittable_list_hit.
 *
 * Record layouts (doubles): ray7 = orig(3) dir(3) time; rec12 = hit, t, p(3), normal(3), u, v,
 * front_face, material type (-1 on miss).
 */
#ifdef HX_ORACLE
#include "cpu_ref.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define HX_DYNAMIC_STAGE 1
#endif

#ifndef RT_SCENES_TYPES
#error "scenes.c must be included/compiled before harness.c (rt_scene_view)"
#endif
hittable_list *rt_scene_build(int id, int arg, hittable_list *lights, rt_scene_view *view);
void rt_scene_set_image_path(const char *path);
void hx_set_image_path(const char *path) { rt_scene_set_image_path(path); }
/* non-static reference functions that its headers do not declare (hittable.c:37, material.c:142) */
void rotatePoint(vec3 original, double sin_theta, double cos_theta, rotation_axis axis, vec3 *output);

static hittable_list *g_world, *g_lights;
static camera *g_cam;
static rt_scene_view g_view;

static int cur_stage(void) {
#ifdef HX_DYNAMIC_STAGE
    return orc_get_stage();
#else
    return HX_STAGE;
#endif
}
int hx_stage(void) { return cur_stage(); }
int hx_set_stage(int s) {
#ifdef HX_DYNAMIC_STAGE
    orc_set_stage(s); return 0;
#else
    return s == HX_STAGE ? 0 : -1;
#endif
}
void hx_srand(unsigned seed) {
    srand(seed);
#ifdef HX_ORACLE
    orc_rng_mode(ORC_RNG_GLIBC);
#endif
}

/* id >= 0: scenes.c; id == -1 / -2: the reference's own simpleScene / setupScene (03 build only),
 * id == -8: the reference's own final_scene (02 build only). */
int hx_build_scene(int id, int arg, double aspect) {
    g_lights = hittable_list_new_no_gc(64);
    memset(&g_view, 0, sizeof g_view);
    if (id >= 0) g_world = rt_scene_build(id, arg, g_lights, &g_view);
#if !defined(HX_ORACLE) && HX_STAGE == 3
    else if (id == -1) { rt_scene_view tmp; g_world = simpleScene(g_lights); hittable_list *l2 = hittable_list_new_no_gc(4); (void)l2; view_cornell(&g_view); g_view.integrator = 0; g_view.max_depth = 10; (void)tmp; }
    else if (id == -2) { g_world = setupScene(g_lights); view_cornell(&g_view); g_view.integrator = 0; g_view.max_depth = 10; }
#endif
#if !defined(HX_ORACLE) && HX_STAGE == 2
    else if (id == -8) { g_world = final_scene(); view_cornell(&g_view); g_view.lookfrom = sv3(478, 278, -600); g_view.integrator = 1; g_view.max_depth = 8; }
    else if (id == -6) { g_world = cornell_box(); view_cornell(&g_view); g_view.integrator = 1; g_view.max_depth = 8; }
#endif
    else g_world = NULL;
    if (!g_world) return -1;
    if (aspect > 0) g_view.aspect = aspect;
    g_cam = camera_new(g_view.lookfrom, g_view.lookat, g_view.vup, g_view.vfov, g_view.aspect, g_view.aperture,
                       g_view.focus_dist, g_view.time0, g_view.time1);
    return 0;
}
void hx_get_view(double *out /*20*/) {
    const rt_scene_view *v = &g_view;
    double tmp[20] = {v->lookfrom.x, v->lookfrom.y, v->lookfrom.z, v->lookat.x, v->lookat.y, v->lookat.z, v->vup.x, v->vup.y, v->vup.z,
                      v->vfov, v->aspect, v->aperture, v->focus_dist, v->time0, v->time1, v->background.x, v->background.y, v->background.z,
                      (double)v->integrator, (double)v->max_depth};
    memcpy(out, tmp, sizeof tmp);
}
void hx_get_camera(double *out /*24*/) {
    const camera *c = g_cam;
    double tmp[24] = {c->origin.x, c->origin.y, c->origin.z, c->lower_left_corner.x, c->lower_left_corner.y, c->lower_left_corner.z,
                      c->horizontal.x, c->horizontal.y, c->horizontal.z, c->vertical.x, c->vertical.y, c->vertical.z,
                      c->u.x, c->u.y, c->u.z, c->v.x, c->v.y, c->v.z, c->w.x, c->w.y, c->w.z, c->lens_radius, c->time0, c->time1};
    memcpy(out, tmp, sizeof tmp);
}

static ray ray_from7(const double *a) { ray r; r.orig.x = a[0]; r.orig.y = a[1]; r.orig.z = a[2]; r.dir.x = a[3]; r.dir.y = a[4]; r.dir.z = a[5]; r.time = a[6]; return r; }
static void ray_to7(const ray *r, double *a) { a[0] = r->orig.x; a[1] = r->orig.y; a[2] = r->orig.z; a[3] = r->dir.x; a[4] = r->dir.y; a[5] = r->dir.z; a[6] = r->time; }
static void rec_to12(int hit, const hit_record *rec, double *o) {
    if (!hit) { for (int i = 0; i < 12; i++) o[i] = 0; o[11] = -1; return; }
    o[0] = 1; o[1] = rec->t; o[2] = rec->p.x; o[3] = rec->p.y; o[4] = rec->p.z;
    o[5] = rec->normal.x; o[6] = rec->normal.y; o[7] = rec->normal.z; o[8] = rec->u; o[9] = rec->v;
    o[10] = rec->front_face; o[11] = (double)((material *)rec->mat)->type;
}

/* camera_get_ray consumes rand() (lens disk + time, F13); call hx_srand first for reproducibility */
int hx_camera_rays(int n, const double *uv, double *rays7) {
    for (int i = 0; i < n; i++) { ray r = camera_get_ray(g_cam, uv[2 * i], uv[2 * i + 1]); ray_to7(&r, rays7 + 7 * i); }
    return 0;
}

/* ---- leaf walker: primitive ids in the reference's visit order ----------------------------- */
/* A leaf is "the" hit leaf when its own single-leaf record has the authoritative t and material.  Where several
 * leaves qualify (coplanar rects of neighbouring boxes share a t), u,v -- which no wrapper rewrites -- tell which one
 * the reference's traversal really kept: its float-reciprocal aabb_hit (F9) may have culled the later candidate.
 * found = last leaf matching t, material, u and v; found_loose = last leaf matching t and material only (used when
 * u,v are NaN, F20). */
typedef struct { double t; void *mat; int next_id; int found; double tmin; double u, v; int found_loose; } leaf_probe;
static void walk(hittable *o, ray r, leaf_probe *pr) {
    switch (o->t) {
    case HITTABLE_SPHERE: case HITTABLE_MOVING_SPHERE: case HITTABLE_RECT: {
        int id = pr->next_id++;
        hit_record rec;
        if (pr->mat && hittable_hit(o, &r, pr->tmin, HUGE_VAL, &rec) && rec.t == pr->t && (void *)rec.mat == pr->mat) {
            pr->found_loose = id;
            if (rec.u == pr->u && rec.v == pr->v) pr->found = id;
        }
        break;
    }
    case HITTABLE_CONSTANT_MEDIUM: { /* stochastic: identified by its private phase-function material */
        int id = pr->next_id++;
        if (pr->mat && (void *)((constant_medium *)o->obj)->phase_function == pr->mat) pr->found = pr->found_loose = id;
        break;
    }
    case HITTABLE_BOX: { hittable_list *l = (hittable_list *)((box *)o->obj)->sides; for (int i = 0; i < l->index; i++) walk(l->objs[i], r, pr); break; }
    case HITTABLE_BVH_NODE: { bvh_node *n = o->obj; walk(n->left, r, pr); if (n->right != n->left) walk(n->right, r, pr); break; }
    case HITTABLE_TRANSLATE: { translate *t = o->obj; r.orig.x -= t->offset.x; r.orig.y -= t->offset.y; r.orig.z -= t->offset.z; walk(t->obj, r, pr); break; }
    case HITTABLE_ROTATE: {
        rotate *ro = o->obj; point3 og = r.orig; vec3 d = r.dir;
        rotatePoint(r.orig, -ro->sin_theta, ro->cos_theta, ro->axis, &og);
        rotatePoint(r.dir, -ro->sin_theta, ro->cos_theta, ro->axis, &d);
        r.orig = og; r.dir = d; walk(ro->obj, r, pr); break;
    }
#if defined(HX_ORACLE) || HX_STAGE == 3
    case HITTABLE_FLIP_FACE: walk(((flip_face *)o->obj)->obj, r, pr); break;
#endif
    default: break;
    }
}
int hx_num_leaves(void) {
    leaf_probe pr = {0, NULL, 0, -1, 0, 0, 0, -1}; ray r; memset(&r, 0, sizeof r); r.dir.x = 1;
    for (int i = 0; i < g_world->index; i++) walk(g_world->objs[i], r, &pr);
    return pr.next_id;
}
int hx_hit_rays(int n, const double *rays7, double tmin, double tmax, double *rec12, int *prim) {
    for (int i = 0; i < n; i++) {
        ray r = ray_from7(rays7 + 7 * i); hit_record rec;
        int hit = hittable_list_hit(g_world, &r, tmin, tmax, &rec);
        rec_to12(hit, &rec, rec12 + 12 * i);
        if (prim) {
            prim[i] = -1;
            if (hit) {
                leaf_probe pr = {rec.t, (void *)rec.mat, 0, -1, tmin, rec.u, rec.v, -1};
                for (int k = 0; k < g_world->index; k++) walk(g_world->objs[k], r, &pr);
                prim[i] = pr.found >= 0 ? pr.found : (pr.found_loose >= 0 ? pr.found_loose : -2); /* -2: walker found no leaf with that t (must not happen) */
            }
        }
    }
    return 0;
}

/* ---- integrator probes ------------------------------------------------------------------- */
static color hx_ray_color(ray *r, int depth) {
#if defined(HX_ORACLE)
    orc_settings st; memset(&st, 0, sizeof st);
    st.integrator = g_view.integrator; st.max_depth = depth; st.mixture_light_weight = 0.1; st.background = g_view.background;
    return orc_ray_color(r, &st, g_world, g_lights, depth);
#elif HX_STAGE == 3
    return ray_color(r, &g_view.background, g_world, g_lights, depth);
#else
    return ray_color(r, &g_view.background, g_world, depth);
#endif
}
/* sequential ray_color over given rays with one rand() stream: pins draw ORDER as well as arithmetic */
int hx_ray_colors(unsigned seed, int n, const double *rays7, int depth, double *rgb) {
    hx_srand(seed);
    for (int i = 0; i < n; i++) { ray r = ray_from7(rays7 + 7 * i); color c = hx_ray_color(&r, depth); rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z; }
    return 0;
}
/* the reference's main loop (03 main.c:132-160 == 02 main.c:411-439), single thread, rand() as is */
int hx_render_refloop(int W, int H, int spp, int depth, unsigned char *pixels) {
    const unsigned THREADS = 4 * 4;
    int step = floor(H / THREADS);
    for (unsigned k = 0; k < THREADS; ++k)
        for (int sj = step - 1; sj >= 0; --sj) {
            int j = sj + (k * step);
            for (int i = 0; i < W; ++i) {
                color pc = {0, 0, 0};
                for (int s = 0; s < spp; ++s) {
                    double u = ((double)i + random_double()) / (double)(W - 1);
                    double v = ((double)j + random_double()) / (double)(H - 1);
                    ray r = camera_get_ray(g_cam, u, v);
                    color c = hx_ray_color(&r, depth);
                    pc.x += c.x; pc.y += c.y; pc.z += c.z;
                }
                write_color(pixels, &pc, spp, (i + (j * W)) * 3);
            }
        }
    return 0;
}

/* the same loop AS SHIPPED: `#pragma omp parallel for` over the 16 row chunks (03 main.c:137, 02 main.c:416), every thread drawing from
 * the one process-wide rand() (utils.h:23; glibc's is lock-protected: more threads are SLOWER).  Timing row only: the
 * image depends on the interleaving of the draws. */
int hx_render_refloop_omp(int W, int H, int spp, int depth, int threads, unsigned char *pixels) {
    const int THREADS = 4 * 4;
    int step = floor(H / THREADS);
    #pragma omp parallel for num_threads(threads)
    for (int k = 0; k < THREADS; ++k)
        for (int sj = step - 1; sj >= 0; --sj) {
            int j = sj + (k * step);
            for (int i = 0; i < W; ++i) {
                color pc = {0, 0, 0};
                for (int s = 0; s < spp; ++s) {
                    double u = ((double)i + random_double()) / (double)(W - 1);
                    double v = ((double)j + random_double()) / (double)(H - 1);
                    ray r = camera_get_ray(g_cam, u, v);
                    color c = hx_ray_color(&r, depth);
                    pc.x += c.x; pc.y += c.y; pc.z += c.z;
                }
                write_color(pixels, &pc, spp, (i + (j * W)) * 3);
            }
        }
    return 0;
}

/* ---- unit probes for pure functions ------------------------------------------------------ */
int hx_unit_rect_hit(const double *rp /*x0,x1,y0,y1,z0,z1,k,axis*/, int n, const double *rays7, double tmin, double tmax, double *rec12) {
    material *m = material_dielectric_new(1.5);
    hittable *h = hittable_rect_new(NULL, rp[0], rp[1], rp[2], rp[3], rp[4], rp[5], rp[6], (rect_axis)(int)rp[7], (struct material *)m);
    for (int i = 0; i < n; i++) { ray r = ray_from7(rays7 + 7 * i); hit_record rec; int hit = hittable_hit(h, &r, tmin, tmax, &rec); rec_to12(hit, &rec, rec12 + 12 * i); }
    return 0;
}
int hx_unit_sphere_hit(const double *sp /*c(3), r*/, int n, const double *rays7, double tmin, double tmax, double *rec12) {
    material *m = material_dielectric_new(1.5);
    point3 c = {sp[0], sp[1], sp[2]};
    hittable *h = hittable_sphere_new(NULL, c, sp[3], (struct material *)m);
    for (int i = 0; i < n; i++) { ray r = ray_from7(rays7 + 7 * i); hit_record rec; int hit = hittable_hit(h, &r, tmin, tmax, &rec); rec_to12(hit, &rec, rec12 + 12 * i); }
    return 0;
}
int hx_unit_moving_sphere_hit(const double *sp /*c0(3), c1(3), t0, t1, r*/, int n, const double *rays7, double tmin, double tmax, double *rec12) {
    material *m = material_dielectric_new(1.5);
    point3 c0 = {sp[0], sp[1], sp[2]}, c1 = {sp[3], sp[4], sp[5]};
    hittable *h = hittable_moving_sphere_new(NULL, c0, c1, sp[6], sp[7], sp[8], (struct material *)m);
    for (int i = 0; i < n; i++) { ray r = ray_from7(rays7 + 7 * i); hit_record rec; int hit = hittable_hit(h, &r, tmin, tmax, &rec); rec_to12(hit, &rec, rec12 + 12 * i); }
    return 0;
}
/* translate(rotate(box)) as in the Cornell block: bp = min(3), max(3), angle, axis, offset(3) */
int hx_unit_box_hit(const double *bp, int n, const double *rays7, double tmin, double tmax, double *rec12) {
    material *m = material_dielectric_new(1.5);
    point3 mn = {bp[0], bp[1], bp[2]}, mx = {bp[3], bp[4], bp[5]}; vec3 off = {bp[8], bp[9], bp[10]};
    hittable *h = hittable_box_new(NULL, mn, mx, (struct material *)m);
    h = hittable_rotate_init(NULL, h, bp[6], (rotation_axis)(int)bp[7]);
    h = hittable_translate_init(NULL, h, off);
    for (int i = 0; i < n; i++) { ray r = ray_from7(rays7 + 7 * i); hit_record rec; int hit = hittable_hit(h, &r, tmin, tmax, &rec); rec_to12(hit, &rec, rec12 + 12 * i); }
    return 0;
}
int hx_unit_aabb_hit(const double *b6, int n, const double *rays7, double tmin, double tmax, int *out) {
    aabb b; b.minimum.x = b6[0]; b.minimum.y = b6[1]; b.minimum.z = b6[2]; b.maximum.x = b6[3]; b.maximum.y = b6[4]; b.maximum.z = b6[5];
    for (int i = 0; i < n; i++) { ray r = ray_from7(rays7 + 7 * i); out[i] = aabb_hit(&b, &r, tmin, tmax); }
    return 0;
}
int hx_unit_refract(int n, const double *in7 /*uv(3) n(3) eta*/, double *out3) {
    for (int i = 0; i < n; i++) {
        const double *a = in7 + 7 * i; vec3 uv = {a[0], a[1], a[2]}, nn = {a[3], a[4], a[5]};
        vec3 r = vec3c_refract(uv, nn, a[6]); out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
    }
    return 0;
}
double reflectance(double cosine, double ref_idx);
int hx_unit_reflectance(int n, const double *in2, double *out) { for (int i = 0; i < n; i++) out[i] = reflectance(in2[2 * i], in2[2 * i + 1]); return 0; }
#if defined(HX_ORACLE) || HX_STAGE == 3
int hx_unit_onb(int n, const double *n3, double *axes9) {
    for (int i = 0; i < n; i++) {
        vec3 w = {n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]}; onb o; onb_init_from_w(&o, &w);
        for (int k = 0; k < 3; k++) { axes9[9 * i + 3 * k] = o.axis[k].x; axes9[9 * i + 3 * k + 1] = o.axis[k].y; axes9[9 * i + 3 * k + 2] = o.axis[k].z; }
    }
    return 0;
}
double pdf_cosine_value(pdf_cosine *p, vec3 *dir);
int hx_unit_cosine_pdf(int n, const double *in6 /*normal(3) dir(3)*/, double *out) {
    for (int i = 0; i < n; i++) {
        vec3 w = {in6[6 * i], in6[6 * i + 1], in6[6 * i + 2]}, d = {in6[6 * i + 3], in6[6 * i + 4], in6[6 * i + 5]};
        pdf_cosine pc; onb_init_from_w(&pc.uvw, &w); out[i] = pdf_cosine_value(&pc, &d);
    }
    return 0;
}
/* list pdf of the CURRENT scene's lights: in6 = origin(3) dir(3) */
int hx_unit_light_pdf(int n, const double *in6, double *out) {
    for (int i = 0; i < n; i++) {
        point3 o = {in6[6 * i], in6[6 * i + 1], in6[6 * i + 2]}; vec3 d = {in6[6 * i + 3], in6[6 * i + 4], in6[6 * i + 5]};
        out[i] = hittable_list_pdf_value(g_lights, &o, &d);
    }
    return 0;
}
int hx_unit_light_random(unsigned seed, int n, const double *orig3, double *dir3) {
    hx_srand(seed);
    for (int i = 0; i < n; i++) {
        point3 o = {orig3[3 * i], orig3[3 * i + 1], orig3[3 * i + 2]}; vec3 d = hittable_list_random(g_lights, o);
        dir3[3 * i] = d.x; dir3[3 * i + 1] = d.y; dir3[3 * i + 2] = d.z;
    }
    return 0;
}
int hx_unit_scattering_pdf(int n, const double *in6 /*normal(3) dir(3)*/, double *out) {
    color c = {0.5, 0.5, 0.5}; material *m = material_lambertian_new_from_color(c);
    for (int i = 0; i < n; i++) {
        hit_record rec; memset(&rec, 0, sizeof rec); rec.normal.x = in6[6 * i]; rec.normal.y = in6[6 * i + 1]; rec.normal.z = in6[6 * i + 2];
        ray s; memset(&s, 0, sizeof s); s.dir.x = in6[6 * i + 3]; s.dir.y = in6[6 * i + 4]; s.dir.z = in6[6 * i + 5];
        out[i] = material_scattering_pdf(m, &s, &rec, &s);
    }
    return 0;
}
#endif
/* material scatter: in14 = ray7, p(3), normal(3), front_face; matp = albedo(3), fuzz | ir.
 * out12 = ret, is_specular(or -1 in stage 2), att(3), scattered ray7 (stage 3 lambertian: zeros) */
int hx_unit_scatter(int mtype, const double *matp, unsigned seed, int n, const double *in14, double *out12) {
    color alb = {matp[0], matp[1], matp[2]};
    material *m = mtype == MATERIAL_LAMBERTIAN ? material_lambertian_new_from_color(alb)
                : mtype == MATERIAL_METAL ? material_metal_new(alb, matp[3])
                : mtype == MATERIAL_DIELECTRIC ? material_dielectric_new(matp[0])
                : mtype == MATERIAL_ISOTROPIC ? material_isotropic_new_from_color(alb) : material_light_new_from_color(alb);
    hx_srand(seed);
    for (int i = 0; i < n; i++) {
        const double *a = in14 + 14 * i; double *o = out12 + 12 * i;
        ray r = ray_from7(a); hit_record rec; memset(&rec, 0, sizeof rec);
        rec.p.x = a[7]; rec.p.y = a[8]; rec.p.z = a[9]; rec.normal.x = a[10]; rec.normal.y = a[11]; rec.normal.z = a[12]; rec.front_face = (int)a[13];
        rec.mat = (struct material *)m;
        for (int k = 0; k < 12; k++) o[k] = 0;
        int stage3 = cur_stage() == 3;
        if (stage3) {
#if defined(HX_ORACLE) || HX_STAGE == 3
            scatter_record srec; memset(&srec, 0, sizeof srec);
            o[0] = material_scatter(m, &r, &rec, &srec);
            if (o[0] != 0 && mtype != MATERIAL_ISOTROPIC) {
                o[1] = srec.is_specular; o[2] = srec.attenuation.x; o[3] = srec.attenuation.y; o[4] = srec.attenuation.z;
                if (srec.is_specular) ray_to7(&srec.specular_ray, o + 5);
                if (srec.pdf_ptr) pdf_free(srec.pdf_ptr);
            }
#endif
        } else {
#if defined(HX_ORACLE) || HX_STAGE == 2
            color att = {0, 0, 0}; ray sc; memset(&sc, 0, sizeof sc);
#if defined(HX_ORACLE)
            o[0] = material_scatter_att(m, &r, &rec, &att, &sc);
#else
            o[0] = material_scatter(m, &r, &rec, &att, &sc);
#endif
            o[1] = -1; o[2] = att.x; o[3] = att.y; o[4] = att.z; ray_to7(&sc, o + 5);
#endif
        }
    }
    return 0;
}
/* rand()-driven samplers, n consecutive draws after srand(seed): pins draw order and arithmetic.
 * kind: 0 vec3_random, 1 in_unit_sphere, 2 unit_vector, 3 in_unit_disk, 4 cosine_direction, 5 random_double (x only) */
vec3 vec3_random(void); vec3 vec3_random_in_unit_sphere(void); vec3 vec3_random_unit_vector_in_unit_sphere(void); vec3 random_in_unit_disk(void);
int hx_unit_sampler(int kind, unsigned seed, int n, double *out3) {
    hx_srand(seed);
    for (int i = 0; i < n; i++) {
        vec3 v = {0, 0, 0};
        switch (kind) {
        case 0: v = vec3_random(); break;
        case 1: v = vec3_random_in_unit_sphere(); break;
        case 2: v = vec3_random_unit_vector_in_unit_sphere(); break;
        case 3: v = random_in_unit_disk(); break;
#if defined(HX_ORACLE) || HX_STAGE == 3
        case 4: v = random_cosine_direction(); break;
#endif
        case 5: v.x = random_double(); break;
        default: return -1;
        }
        out3[3 * i] = v.x; out3[3 * i + 1] = v.y; out3[3 * i + 2] = v.z;
    }
    return 0;
}
/* perlin: tables after srand(seed), then noise/turb at the given points */
int hx_unit_perlin(unsigned seed, int n, const double *p3, double *noise, double *turb, double *ranvec768, int *perm768) {
    hx_srand(seed);
    perlin *p = perlin_init();
    if (ranvec768) for (int i = 0; i < 256; i++) { ranvec768[3 * i] = p->ranvec[i].x; ranvec768[3 * i + 1] = p->ranvec[i].y; ranvec768[3 * i + 2] = p->ranvec[i].z; }
    if (perm768) for (int i = 0; i < 256; i++) { perm768[i] = p->perm_x[i]; perm768[256 + i] = p->perm_y[i]; perm768[512 + i] = p->perm_z[i]; }
    for (int i = 0; i < n; i++) { point3 q = {p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]}; noise[i] = perlin_noise(p, q); turb[i] = perlin_turb(p, q); }
    return 0;
}
/* textures: kind 0 solid(c0), 1 checker(c0 odd, c1 even), 2 noise(scale=c0[0], srand(seed)); in5 = u, v, p(3) */
int hx_unit_texture(int kind, const double *c0, const double *c1, unsigned seed, int n, const double *in5, double *out3) {
    texture *t;
    color a = {c0[0], c0[1], c0[2]}, b = {c1[0], c1[1], c1[2]};
    if (kind == 0) t = texture_solid_color_init(a);
    else if (kind == 1) t = texture_checker_init_c(a, b);
    else { hx_srand(seed); t = texture_noise_init_scaled(c0[0]); }
    for (int i = 0; i < n; i++) {
        vec3 p = {in5[5 * i + 2], in5[5 * i + 3], in5[5 * i + 4]};
        color c = texture_value(t, in5[5 * i], in5[5 * i + 1], &p); out3[3 * i] = c.x; out3[3 * i + 1] = c.y; out3[3 * i + 2] = c.z;
    }
    return 0;
}
int hx_unit_write_color(int n, const double *rgb3, int spp, unsigned char *out3) {
    for (int i = 0; i < n; i++) { color c = {rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]}; write_color(out3, &c, spp, 3 * i); }
    return 0;
}
/* dump of the flattened leaves, for cross-checking scene construction between backends:
 * per leaf 16 doubles: type, mat type, geometry (up to 10), n wrappers above, flipped */
static void dump_walk(hittable *o, int depth, int flipped, double *out, int *n, int cap) {
    switch (o->t) {
    case HITTABLE_SPHERE: { sphere *s = o->obj; if (*n < cap) { double *d = out + 16 * *n; memset(d, 0, 16 * sizeof(double)); d[0] = o->t; d[1] = ((material *)s->mat)->type; d[2] = s->center.x; d[3] = s->center.y; d[4] = s->center.z; d[5] = s->r; d[12] = depth; d[13] = flipped; } (*n)++; break; }
    case HITTABLE_MOVING_SPHERE: { moving_sphere *s = o->obj; if (*n < cap) { double *d = out + 16 * *n; memset(d, 0, 16 * sizeof(double)); d[0] = o->t; d[1] = ((material *)s->mat)->type; d[2] = s->center0.x; d[3] = s->center0.y; d[4] = s->center0.z; d[5] = s->center1.x; d[6] = s->center1.y; d[7] = s->center1.z; d[8] = s->time0; d[9] = s->time1; d[10] = s->r; d[12] = depth; d[13] = flipped; } (*n)++; break; }
    case HITTABLE_RECT: { rect *s = o->obj; if (*n < cap) { double *d = out + 16 * *n; memset(d, 0, 16 * sizeof(double)); d[0] = o->t; d[1] = ((material *)s->mat)->type; d[2] = s->x0; d[3] = s->x1; d[4] = s->y0; d[5] = s->y1; d[6] = s->z0; d[7] = s->z1; d[8] = s->k; d[9] = s->axis; d[12] = depth; d[13] = flipped; } (*n)++; break; }
    case HITTABLE_CONSTANT_MEDIUM: { constant_medium *m = o->obj; if (*n < cap) { double *d = out + 16 * *n; memset(d, 0, 16 * sizeof(double)); d[0] = o->t; d[1] = MATERIAL_ISOTROPIC; d[2] = m->neg_inv_density; d[12] = depth; d[13] = flipped; } (*n)++; break; }
    case HITTABLE_BOX: { hittable_list *l = (hittable_list *)((box *)o->obj)->sides; for (int i = 0; i < l->index; i++) dump_walk(l->objs[i], depth, flipped, out, n, cap); break; }
    case HITTABLE_BVH_NODE: { bvh_node *b = o->obj; dump_walk(b->left, depth, flipped, out, n, cap); if (b->right != b->left) dump_walk(b->right, depth, flipped, out, n, cap); break; }
    case HITTABLE_TRANSLATE: dump_walk(((translate *)o->obj)->obj, depth + 1, flipped, out, n, cap); break;
    case HITTABLE_ROTATE: dump_walk(((rotate *)o->obj)->obj, depth + 1, flipped, out, n, cap); break;
#if defined(HX_ORACLE) || HX_STAGE == 3
    case HITTABLE_FLIP_FACE: dump_walk(((flip_face *)o->obj)->obj, depth, !flipped, out, n, cap); break;
#endif
    default: break;
    }
}
int hx_dump_leaves(double *out16, int cap) { int n = 0; for (int i = 0; i < g_world->index; i++) dump_walk(g_world->objs[i], 0, 0, out16, &n, cap); return n; }

#ifdef HX_ORACLE
int hx_counters(int reset, unsigned long long *out, int cap) {
    if (reset) { orc_counters_reset(); return 0; }
    orc_counters c; orc_counters_get(&c);
    int n = (int)(sizeof c / sizeof(unsigned long long)); if (n > cap) n = cap;
    memcpy(out, &c, (size_t)n * sizeof(unsigned long long)); return n;
}
/* ---- oracle-only conveniences (scene state lives in this file) ---------------------------- */
static orc_settings settings_from_view(int depth, int scrub) {
    orc_settings st; memset(&st, 0, sizeof st);
    st.integrator = g_view.integrator; st.max_depth = depth > 0 ? depth : g_view.max_depth; st.mixture_light_weight = 0.1;
    st.background = g_view.background; st.scrub_nan = scrub;
    return st;
}
/* multi-threaded Philox (or single-thread rand()) render of the current scene into a double RGB sum */
int hx_oracle_render(int W, int H, int s_begin, int s_end, int depth, int rng_mode, unsigned long long seed, int threads,
                     int iterative, int scrub_nan, double *rgb_sum, double *stats3 /*samples, rays, seconds*/) {
    orc_settings st = settings_from_view(depth, scrub_nan);
    orc_render_stats rs;
    int rc = orc_render(g_world, g_lights, g_cam, &st, W, H, s_begin, s_end, rng_mode, seed, threads, iterative, rgb_sum, &rs);
    if (stats3) { stats3[0] = (double)rs.samples; stats3[1] = (double)rs.rays; stats3[2] = rs.seconds; }
    return rc;
}
/* 01's output edge (stage 1): the serial loop of 01 main.c:168-191 into an ASCII PPM; reseed = 0 continues the rand()
 * stream where scene construction left it, as the reference's main() does */
int hx_render_ppm01(const char *path, int W, int H, int spp, int depth, int reseed, unsigned seed) {
    if (reseed) hx_srand(seed);
    orc_settings st = settings_from_view(depth, 0);
    return orc_render_ppm01(g_world, g_cam, &st, W, H, spp, path);
}
int hx_write_colors_ppm(const char *path, int n, const double *rgb3, int spp) {
    FILE *f = fopen(path, "w+");
    if (!f) return -1;
    for (int i = 0; i < n; i++) { color c = {rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]}; orc_write_color_ppm(f, &c, spp); }
    fclose(f);
    return 0;
}
int hx_ray_colors_iter(unsigned seed, int n, const double *rays7, int depth, double *rgb) {
    hx_srand(seed);
    orc_settings st = settings_from_view(depth, 0);
    for (int i = 0; i < n; i++) { ray r = ray_from7(rays7 + 7 * i); color c = orc_ray_color_iter(&r, &st, g_world, g_lights, NULL); rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z; }
    return 0;
}
#endif
