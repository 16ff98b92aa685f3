/*
 * facade.c -- make_scene / render_scene / free_scene (product code, host side).
 *
 * The reference's library build (`make lib`, Makefile:55-62) exposes its renderer to host languages through three
 * calls (src/renderer.h:56-63): make_scene(objects, lights, size), render_scene(scene, camera, settings, filename),
 * free_scene(scene), with scene_settings = {res_width, res_height, samples_per_pixel, iterations} (renderer.h:31-36)
 * where a render is `iterations` passes of `samples_per_pixel` samples each (renderer.c:372-375) and ends in a PNG
 * (renderer.c:429).  This file keeps those three names and that struct over the C-ABI of include/rtc.h; the passes
 * accumulate into one float4 image, which is what makes a render resumable.  Scene conventions are the 03 tracer's:
 * black background (03 main.c:119), shutter 0..1, mixture-PDF integrator at depth 10 (ray.c:50-84, main.c:92).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtc.h"

scene *make_scene(const hittable_list *objects, const hittable_list *lights, int size) {
    (void)size;                                     /* the reference's table capacity; tables here grow as needed */
    if (rtc_init() != RTC_OK) return NULL;
    const color black = {0, 0, 0};
    return rtc_scene_build(objects, lights, &black, 0.0, 1.0);
}

void free_scene(scene *s) { rtc_scene_free(s); }

int render_scene_accumulate(const scene *s, const camera *cam, const scene_settings *settings, int first_iteration, int n_iterations,
                            float *accum_rgba_host) {
    if (!s || !cam || !settings || !accum_rgba_host || first_iteration < 0 || n_iterations < 0) return RTC_ERR_ARG;
    rtc_settings st; memset(&st, 0, sizeof st);
    st.width = settings->res_width; st.height = settings->res_height; st.max_depth = 10; st.integrator = 0;
    st.seed = 0; st.mixture_light_weight = 0.1;
    for (int i = first_iteration; i < first_iteration + n_iterations; i++) {
        st.sample_begin = i * settings->samples_per_pixel; st.sample_end = (i + 1) * settings->samples_per_pixel;
        const int rc = rtc_render(s, cam, &st, accum_rgba_host);
        if (rc != RTC_OK) return rc;
    }
    return RTC_OK;
}

void render_scene(const scene *s, const camera *cam, const scene_settings *settings, const char *filename) {
    if (!s || !cam || !settings || !filename) { fprintf(stderr, "render_scene: null argument\n"); return; }
    const size_t n = (size_t)settings->res_width * (size_t)settings->res_height;
    float *accum = (float *)calloc(n * 4, sizeof(float));
    unsigned char *pixels = (unsigned char *)malloc(n * 3 + 1);
    if (!accum || !pixels) { fprintf(stderr, "render_scene: out of memory\n"); free(accum); free(pixels); return; }
    if (render_scene_accumulate(s, cam, settings, 0, settings->iterations, accum) != RTC_OK ||
        rtc_tonemap_u8(accum, settings->res_width, settings->res_height, pixels) != RTC_OK)
        fprintf(stderr, "render_scene: %s\n", rtc_last_error());
    else if (!rt_write_png(filename, settings->res_width, settings->res_height, pixels, /*flip_vertically=*/1))
        fprintf(stderr, "render_scene: cannot write %s\n", filename);
    free(accum); free(pixels);
}
