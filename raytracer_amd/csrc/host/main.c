/*
 * main.c -- command-line front end of raytracer_amd (product code).
 * Keeps the reference's flags (03_FullTracer/main.c:93-105: -w width, -h height, -o outputFile, -s samples) and
 * its PNG output (vertical flip, main.c:170-171); the OpenMP pixel loop (main.c:137-167) is replaced by
 * rtc_render on the GPU.  Extra flags: -S scene id (default 1 = Cornell box; the reference hard-codes the scene),
 * -a aspect (default W/H; the reference hard-codes 1.0), -d depth, -r seed, -A arg (scene size),
 * -W file (write the flattened scene, rtc_scene_save), -L file (render that file instead of flattening scene -S again;
 * -S still names the camera), -g N (render on N GPUs of this process: rtc_render_multi, sample shards, peer-copy reduce;
 * 0 = all visible devices).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rtc.h"

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int main(int argc, char **argv) {
    const char *out = "output.png", *save_path = NULL, *load_path = NULL;
    int W = 1024, H = -1, spp = 256, depth = -1, scene_id = RT_SCENE_CORNELL, arg = 0, n_gpus = 1;
    double aspect = -1; unsigned long long seed = 0;
    for (int i = 1; i < argc; i++) {
        if (argv[i][0] != '-') continue;
        if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", argv[i]); return EXIT_FAILURE; }
        switch (argv[i][1]) {
        case 'w': W = atoi(argv[i + 1]); break;
        case 'h': H = atoi(argv[i + 1]); break;
        case 'o': out = argv[i + 1]; break;
        case 's': spp = atoi(argv[i + 1]); break;
        case 'S': scene_id = atoi(argv[i + 1]); break;
        case 'a': aspect = atof(argv[i + 1]); break;
        case 'd': depth = atoi(argv[i + 1]); break;
        case 'r': seed = strtoull(argv[i + 1], NULL, 0); break;
        case 'A': arg = atoi(argv[i + 1]); break;
        case 'W': save_path = argv[i + 1]; break;
        case 'L': load_path = argv[i + 1]; break;
        case 'g': n_gpus = atoi(argv[i + 1]); break;
        default:
            fprintf(stderr, "Usage: %s [-w width] [-h height] [-o outputFile] [-s sample_per_pixel] [-S scene] [-a aspect] [-d depth] [-r seed] [-A scene_arg] [-W scene_file_out] [-L scene_file_in] [-g gpus]\n", argv[0]);
            return EXIT_FAILURE;
        }
        i++;
    }
    if (rtc_init() != RTC_OK) { fprintf(stderr, "rtc_init: %s\n", rtc_last_error()); return EXIT_FAILURE; }
    hittable_list *lights = hittable_list_new_no_gc(64);
    rt_scene_view view; memset(&view, 0, sizeof view);
    srand(1);
    hittable_list *world = rt_scene_build(scene_id, arg, lights, &view);
    if (!world) { fprintf(stderr, "unknown scene %d\n", scene_id); return EXIT_FAILURE; }
    if (H <= 0) H = (int)(W / (aspect > 0 ? aspect : view.aspect));
    if (aspect <= 0) aspect = (double)W / (double)H;
    if (depth < 0) depth = view.max_depth;
    camera *cam = camera_new(view.lookfrom, view.lookat, view.vup, view.vfov, aspect, view.aperture, view.focus_dist, view.time0, view.time1);

    double t0 = now_s();
    rtc_scene *scene = load_path ? rtc_scene_load(load_path) : rtc_scene_build(world, lights, &view.background, view.time0, view.time1);
    if (!scene) { fprintf(stderr, "%s\n", rtc_last_error()); return EXIT_FAILURE; }
    if (save_path && rtc_scene_save(scene, save_path) != RTC_OK) { fprintf(stderr, "%s\n", rtc_last_error()); return EXIT_FAILURE; }
    double t1 = now_s();
    rtc_settings st; memset(&st, 0, sizeof st);
    st.width = W; st.height = H; st.max_depth = depth; st.sample_begin = 0; st.sample_end = spp;
    st.integrator = view.integrator; st.seed = seed; st.mixture_light_weight = 0.1;
    float *accum = (float *)calloc((size_t)W * H * 4, sizeof(float));
    unsigned char *pixels = (unsigned char *)malloc((size_t)W * H * 3);
    rtc_stats stats; memset(&stats, 0, sizeof stats);
    enum { MAX_GPUS = 64 };
    rtc_scene *scenes[MAX_GPUS] = {scene};
    if (n_gpus <= 0) n_gpus = rtc_device_count();
    if (n_gpus > MAX_GPUS || n_gpus > rtc_device_count()) { fprintf(stderr, "-g %d: this process sees %d device(s)\n", n_gpus, rtc_device_count()); return EXIT_FAILURE; }
    if (n_gpus <= 1) {
        if (rtc_render(scene, cam, &st, accum) != RTC_OK) { fprintf(stderr, "rtc_render: %s\n", rtc_last_error()); return EXIT_FAILURE; }
        rtc_render_stats(scene, NULL, &stats);
    } else {   /* one host thread and one copy of the tables per device; the flattening above is not repeated */
        for (int d = 1; d < n_gpus; d++)
            if (!(scenes[d] = rtc_scene_clone_to_device(scene, d))) { fprintf(stderr, "%s\n", rtc_last_error()); return EXIT_FAILURE; }
        if (rtc_render_multi((const rtc_scene *const *)scenes, n_gpus, cam, &st, accum) != RTC_OK) { fprintf(stderr, "rtc_render_multi: %s\n", rtc_last_error()); return EXIT_FAILURE; }
        for (int d = 0; d < n_gpus; d++) {   /* samples and rays add up; the devices ran side by side: the slowest one is the kernel time */
            rtc_stats one; rtc_render_stats(scenes[d], NULL, &one);
            stats.samples += one.samples; stats.rays += one.rays;
            if (one.kernel_ms > stats.kernel_ms) stats.kernel_ms = one.kernel_ms;
        }
    }
    double t2 = now_s();
    if (rtc_tonemap_u8(accum, W, H, pixels) != RTC_OK) { fprintf(stderr, "rtc_tonemap_u8: %s\n", rtc_last_error()); return EXIT_FAILURE; }
    if (!rt_write_png(out, W, H, pixels, 1)) { fprintf(stderr, "could not write %s\n", out); return EXIT_FAILURE; }
    double t3 = now_s();
    printf("%s: %d prims, scene %s %.3f s\n", rtc_device_name(), rtc_scene_num_prims(scene), load_path ? "load" : "build", t1 - t0);
    printf("Time elpased for rendering %f\n", t2 - t1);
    printf("kernel %.3f ms: %.1f Msamples/s, %.1f Mrays/s (%.2f rays/sample); png %.3f s -> %s\n", stats.kernel_ms,
           stats.samples / stats.kernel_ms * 1e-3, stats.rays / stats.kernel_ms * 1e-3, (double)stats.rays / (double)stats.samples, t3 - t2, out);

    for (int d = 1; d < n_gpus; d++) rtc_scene_free(scenes[d]);
    rtc_scene_free(scene); free(accum); free(pixels);
    camera_free(cam);
    hittable_lists_free_all(); materials_free_all(); textures_free_all();
    return 0;
}
