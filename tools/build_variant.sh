#!/bin/bash
# Build an experimental variant of the engine: tools/build_variant.sh NAME [extra hipcc flags...]  ->  build_variants/libfb_NAME.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p "$R/build_variants"
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DFB_BUILD_ID="\"variant-$N\"" "$@" \
  -o "$R/build_variants/libfb_$N.so" "$R/flybody_amd/csrc/fb_engine.hip"
echo "built build_variants/libfb_$N.so"
