#!/bin/bash
# tools/exp/shade_grid.sh -- k_wf_shade's footprint: workgroup size (library build) x number of persistent workgroups (RTC_WF_SHADE_WGS), one GPU run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for pass in 1 2; do for v in "librtc.so 256" "librtc.so 128" "librtc.so 64" "librtc_s256.so 1024" "librtc_s256.so 512" "librtc_s256.so 256" "librtc_s256.so 128"; do set -- $v
  echo "== $1 wgs=$2 (pass $pass)"; RTC_LIB=$ROOT/raytracer_amd/$1 RTC_WF_SHADE_WGS=$2 timeout -k 10 300 python3 "$ROOT/tools/bench_configs.py" final_1024 final_earth_1080p 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-18s %9.3f ms' % (d['config'], d['kernel_ms']))" || exit 1; done; done
