#!/bin/bash
# tools/exp/shade_grid.sh -- k_wf_shade's footprint: workgroup size and register budget (library builds) x number of persistent workgroups
# (RTC_WF_SHADE_WGS; 0 = the default, 1024 threads per CU), one GPU run, two passes.  VARIANTS="lib:wgs ..."
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for pass in 1 2; do for v in $VARIANTS; do L=${v%%:*}; W=${v##*:}
  echo "== $L wgs=$W (pass $pass)"; RTC_LIB=$ROOT/raytracer_amd/$L RTC_WF_SHADE_WGS=$W timeout -k 10 300 python3 "$ROOT/tools/bench_configs.py" "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-18s %9.3f ms' % (d['config'], d['kernel_ms']))" || exit 1; done; done
