#!/bin/bash
# Profiling build of the engine (per-phase s_memtime counters, FB_PROF field): used by tools/phase_profile.py and
# tools/tail_profile.py only, never by the package.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
FLAGS=$(sed -n 's#^// FB_HIPCC_FLAGS:##p' "$R/flybody_amd/csrc/fb_build_flags.h")      # the package's own extra flags (FB_NO_BASE_FLAGS=1: without them)
[ -n "$FB_NO_BASE_FLAGS" ] && FLAGS=""
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $FLAGS -DFB_PROFILE \
  -o "$R/flybody_amd/libflybody_hip_prof.so" "$R/flybody_amd/csrc/fb_engine.hip"
echo "built $R/flybody_amd/libflybody_hip_prof.so"
