/* oracle/ref_shim/ref01_tu.c -- TEST INFRASTRUCTURE.
 * The reference's 01_BasicTracer/src/main.c (a header-only single translation unit, compiled where it lies under
 * /root/reference, its main renamed) followed by the 01 probe harness.  Nothing from /root/reference is copied. */
#define main ref01_main
#include "main.c"          /* -I/root/reference/01_BasicTracer/src */
#undef main
#include "../harness01.c"
