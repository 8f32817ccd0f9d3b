#!/bin/bash
# Build a variant of the library that differs from the product in ONE source file's -D flags (A/B runs with tools/ab_lib.sh / ab_lib_cmd.sh):
#   tools/build_variant.sh NAME FILE.hip -DFOO=1 [...]  ->  tools/_ab/libace355_NAME.so   (the product library is (re)built first)
# The variant directory travels with gpurun pushes (the .so files are git-ignored); delete what a measurement no longer needs.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
NAME=$1; FILE=$2; shift 2
python -c "from ace355 import build; build.build()"
C="$ROOT/ace-step-1.5-for-windows_amd/csrc"
OUT="$ROOT/tools/_ab"
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT"
EXTRA=""
if [ "$FILE" = "gemm.hip" ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA "$@" -c "$C/$FILE" -o "$TMP/${NAME}.o" 2> "$TMP/log" || { grep -E "error" "$TMP/log"; exit 1; }
OBJS=""
for f in gemm attn elementwise conv dit vae cond audio_out api; do
  if [ "$f.hip" = "$FILE" ]; then OBJS="$OBJS $TMP/${NAME}.o"; else OBJS="$OBJS $C/_build/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/libace355_${NAME}.so"
ls -la "$OUT/libace355_${NAME}.so"
