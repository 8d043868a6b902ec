#!/bin/bash
# tools/_dev_build_variant.sh <out.so> <rows> <cols> [-D...]: one-geometry build of the library for in-process A/B runs (tools/lib_ab.py)
OUT=$1; R=$2; C=$3; shift 3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -DSGX_BUILD_ID=\"variant\" -DSGX_ONLY_EXTRA -DSGX_EXTRA_R=$R -DSGX_EXTRA_C=$C "$@" \
  -I $ROOT/include ${SGX_SRC:-$ROOT/stratego_env_amd/csrc}/stratego_mi355x.hip -o $OUT 2>&1 | grep -v "warning:" | head -5
