/* oracle/ref_shim/ref02_tu.c -- TEST INFRASTRUCTURE.
 * The reference's 02_AdvancedTracer/src/main.c (where it lies, main renamed) + scenes + harness. */
#define main ref02_main
#include "main.c"          /* -I/root/reference/02_AdvancedTracer/src */
#undef main
#define HX_STAGE 2
#define RT_NO_FLIP_FACE 1
#include "../../raytracer_amd/csrc/host/scenes.c"
#include "../harness.c"
