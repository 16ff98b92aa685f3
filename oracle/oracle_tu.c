/* oracle/oracle_tu.c -- TEST INFRASTRUCTURE: scenes + harness compiled against the CPU oracle. */
#define HX_ORACLE 1
#define RT_API_HEADER "cpu_ref.h"
#include "../raytracer_amd/csrc/host/scenes.c"
#include "harness.c"
