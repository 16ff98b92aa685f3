/*
 * oracle/harness01.c -- flat, ctypes-friendly probes of the reference's 01_BasicTracer (TEST INFRASTRUCTURE).
 *
 * Compiled ONLY textually after the reference's 01_BasicTracer/src/main.c (ref_shim/ref01_tu.c -> oracle/_ref/libref01.so):
 * 01 is a header-only single translation unit with its own API (hittable_sphere_new(world, x, y, z, r, mat), a ray without
 * time, a hit_record without u,v, camera_new without a shutter, write_color(FILE*, ...)), so the 02/03 harness does not
 * fit it.  Every function below calls the reference's OWN setup_scene / camera_get_ray / hittable_list_hit / ray_color /
 * write_color; the only logic of its own is the sample loop of 01 main.c:168-191 (main() cannot be called at another
 * size: its constants are locals), and hx01_run_main runs the real main() so that the generator can check that the two
 * agree byte for byte at the size main() is hard-wired to.
 *
 * Records (doubles): ray6 = orig(3) dir(3); rec9 = hit, t, p(3), normal(3), front_face; sphere10 = centre(3), r,
 * material type (01 material.h:13), albedo(3) | ir, fuzz.
 */
static hittable_list *g01_world;
static camera *g01_cam;

int hx01_build(unsigned seed, double aspect) {
    srand(seed);
    g01_world = setup_scene();                                                       /* 01 main.c:72-116 */
    point3 lookfrom = {13, 2, 3}, lookat = {0, 0, 0}; vec3 vup = {0, 1, 0};
    g01_cam = camera_new(lookfrom, lookat, vup, 30, aspect > 0 ? aspect : 16.0 / 9.0, 0.1, 10);   /* 01 main.c:142-148 */
    return g01_world->index;
}
void hx01_srand(unsigned seed) { srand(seed); }
void hx01_get_camera(double *out /*22*/) {
    const camera *c = g01_cam;
    double tmp[22] = {c->origin.x, c->origin.y, c->origin.z, c->lower_left_corner.x, c->lower_left_corner.y, c->lower_left_corner.z,
                      c->horizontal.x, c->horizontal.y, c->horizontal.z, c->vertical.x, c->vertical.y, c->vertical.z,
                      c->u.x, c->u.y, c->u.z, c->v.x, c->v.y, c->v.z, c->w.x, c->w.y, c->w.z, c->lens_radius};
    for (int i = 0; i < 22; i++) out[i] = tmp[i];
}
int hx01_dump_spheres(double *out10, int cap) {
    for (int i = 0; i < g01_world->index && i < cap; i++) {
        const sphere *s = g01_world->objs[i]->obj; const material *m = (const material *)s->mat; double *d = out10 + 10 * i;
        for (int k = 0; k < 10; k++) d[k] = 0;
        d[0] = s->center.x; d[1] = s->center.y; d[2] = s->center.z; d[3] = s->r; d[4] = m->type;
        if (m->type == MATERIAL_LAMBERTIAN) { const material_lambertian *l = m->mat; d[5] = l->albedo.x; d[6] = l->albedo.y; d[7] = l->albedo.z; }
        else if (m->type == MATERIAL_METAL) { const material_metal *l = m->mat; d[5] = l->albedo.x; d[6] = l->albedo.y; d[7] = l->albedo.z; d[8] = l->fuzz; }
        else { const material_dielectric *l = m->mat; d[5] = l->ir; }
    }
    return g01_world->index;
}
static ray ray_from6(const double *a) { ray r = {{a[0], a[1], a[2]}, {a[3], a[4], a[5]}}; return r; }
/* camera_get_ray draws the lens disk from rand() (01 camera.h:75-82): hx01_srand first */
int hx01_camera_rays(int n, const double *uv, double *rays6) {
    for (int i = 0; i < n; i++) {
        ray r = camera_get_ray(g01_cam, uv[2 * i], uv[2 * i + 1]); double *o = rays6 + 6 * i;
        o[0] = r.orig.x; o[1] = r.orig.y; o[2] = r.orig.z; o[3] = r.dir.x; o[4] = r.dir.y; o[5] = r.dir.z;
    }
    return 0;
}
/* prim = index of the sphere in the world list (the list scan keeps the LAST sphere among equal t, 01 hittable_list.h:40-50) */
int hx01_hit_rays(int n, const double *rays6, double tmin, double tmax, double *rec9, int *mat_type, int *prim) {
    for (int i = 0; i < n; i++) {
        ray r = ray_from6(rays6 + 6 * i); hit_record rec; double *o = rec9 + 9 * i;
        int hit = hittable_list_hit(g01_world, &r, tmin, tmax, &rec);
        for (int k = 0; k < 9; k++) o[k] = 0;
        mat_type[i] = -1; prim[i] = -1;
        if (!hit) continue;
        o[0] = 1; o[1] = rec.t; o[2] = rec.p.x; o[3] = rec.p.y; o[4] = rec.p.z; o[5] = rec.normal.x; o[6] = rec.normal.y; o[7] = rec.normal.z; o[8] = rec.front_face;
        mat_type[i] = ((material *)rec.mat)->type;
        for (int k = 0; k < g01_world->index; k++) {
            hit_record r2;
            if (hittable_hit(g01_world->objs[k], &r, tmin, tmax, &r2) && r2.t == rec.t && r2.mat == rec.mat) prim[i] = k;
        }
    }
    return 0;
}
/* sequential ray_color over given rays with one rand() stream: pins the draw ORDER as well as the arithmetic */
int hx01_ray_colors(unsigned seed, int n, const double *rays6, int depth, double *rgb) {
    srand(seed);
    for (int i = 0; i < n; i++) { ray r = ray_from6(rays6 + 6 * i); color c = ray_color(&r, g01_world, depth); rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z; }
    return 0;
}
/* material_scatter for given hit records: in13 = ray6, p(3), normal(3), front_face; matp = albedo(3), fuzz | ir;
 * out10 = ret, attenuation(3), scattered ray6 */
int hx01_scatter(int mtype, const double *matp, unsigned seed, int n, const double *in13, double *out10) {
    color alb = {matp[0], matp[1], matp[2]};
    material *m = mtype == MATERIAL_LAMBERTIAN ? material_lambertian_new(alb) : mtype == MATERIAL_METAL ? material_metal_new(alb, matp[3]) : material_dielectric_new(matp[0]);
    srand(seed);
    for (int i = 0; i < n; i++) {
        const double *a = in13 + 13 * i; double *o = out10 + 10 * i;
        ray r = ray_from6(a); hit_record rec;
        rec.p = (point3){a[6], a[7], a[8]}; rec.normal = (vec3){a[9], a[10], a[11]}; rec.front_face = (int)a[12]; rec.mat = (struct material *)m; rec.t = 0;
        color att = {0, 0, 0}; ray sc = {{0, 0, 0}, {0, 0, 0}};
        o[0] = material_scatter(m, &r, &rec, &att, &sc);
        o[1] = att.x; o[2] = att.y; o[3] = att.z; o[4] = sc.orig.x; o[5] = sc.orig.y; o[6] = sc.orig.z; o[7] = sc.dir.x; o[8] = sc.dir.y; o[9] = sc.dir.z;
    }
    return 0;
}
/* the render loop of 01 main.c:168-191 at a chosen size, written through the reference's own write_color(FILE*, ...) */
int hx01_render_ppm(const char *path, int W, int H, int spp, int depth, int reseed, unsigned seed) {
    FILE *f = fopen(path, "w+");
    if (!f) return -1;
    if (reseed) srand(seed);                                       /* 0: the stream continues from scene construction, as in main() */
    fprintf(f, "P3\n%d %d\n255\n", W, H);
    for (int j = H - 1; j >= 0; --j)
        for (int i = 0; i < W; ++i) {
            color pixel_color = {0, 0, 0};
            for (int s = 0; s < spp; ++s) {
                double u = ((double)i + random_double()) / (double)(W - 1);
                double v = ((double)j + random_double()) / (double)(H - 1);
                ray r = camera_get_ray(g01_cam, u, v);
                pixel_color = vec3c_sum(pixel_color, ray_color(&r, g01_world, depth));
            }
            write_color(f, &pixel_color, spp);
        }
    fclose(f);
    return 0;
}
/* write_color alone: n colour sums -> the text it prints */
int hx01_write_colors(const char *path, int n, const double *rgb3, int spp) {
    FILE *f = fopen(path, "w+");
    if (!f) return -1;
    for (int i = 0; i < n; i++) { color c = {rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]}; write_color(f, &c, spp); }
    fclose(f);
    return 0;
}
/* the reference's real main(): srand is never called there (seed 1), scene + 1920x1080 x 8 spp x depth 8, about 2 minutes */
int hx01_run_main(const char *path) {
    char a0[] = "ref01"; char *argv[3] = {a0, (char *)path, NULL};
    srand(1);
    return ref01_main(2, argv);
}
