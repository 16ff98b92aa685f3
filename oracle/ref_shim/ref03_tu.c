/* oracle/ref_shim/ref03_tu.c -- TEST INFRASTRUCTURE.
 * One translation unit = the reference's 03_FullTracer/main.c (compiled where it lies under
 * /root/reference, its main renamed) followed by this repo's scene file and probe harness, so the
 * harness calls the reference's own functions.  Nothing from /root/reference is copied into the repo. */
#define main ref03_main
#include "main.c"          /* -I/root/reference/03_FullTracer */
#undef main
#define HX_STAGE 3
#include "../../raytracer_amd/csrc/host/scenes.c"
#include "../harness.c"
