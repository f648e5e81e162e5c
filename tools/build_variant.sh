#!/bin/bash
# Build an experimental variant of the engine: tools/build_variant.sh NAME [extra hipcc flags...]  ->  build_variants/libfb_NAME.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p "$R/build_variants"
FLAGS=$(sed -n 's#^// FB_HIPCC_FLAGS:##p' "$R/flybody_amd/csrc/fb_build_flags.h")      # the package's own extra flags (FB_NO_BASE_FLAGS=1: without them)
[ -n "$FB_NO_BASE_FLAGS" ] && FLAGS=""
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $FLAGS -DFB_BUILD_ID="\"variant-$N\"" "$@" \
  -o "$R/build_variants/libfb_$N.so" "$R/flybody_amd/csrc/fb_engine.hip"
echo "built build_variants/libfb_$N.so"
