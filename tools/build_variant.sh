#!/bin/bash
# Build a variant of the library that differs from the product in ONE source file's -D flags (A/B runs with tools/ab_lib.sh):
#   tools/build_variant.sh NAME FILE.hip -DFOO=1 [...]  ->  csrc/_variants/libace355_NAME.so   (the product library is (re)built first)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
NAME=$1; FILE=$2; shift 2
python -c "from ace355 import build; build.build()"
C="$ROOT/ace-step-1.5-for-windows_amd/csrc"
mkdir -p $C/_variants /tmp/var
EXTRA=""
if [ "$FILE" = "gemm.hip" ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA "$@" -c $C/$FILE -o /tmp/var/${NAME}.o 2>&1 | grep -E "error" || true
OBJS=""
for f in gemm attn elementwise conv dit vae cond audio_out api; do
  if [ "$f.hip" = "$FILE" ]; then OBJS="$OBJS /tmp/var/${NAME}.o"; else OBJS="$OBJS $C/_build/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $C/_variants/libace355_${NAME}.so
ls -la $C/_variants/libace355_${NAME}.so
