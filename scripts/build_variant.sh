#!/bin/bash
# Developer tool: build a variant of librmhip.so with extra -D flags for ONE translation unit (default lu.hip) into
# ab_old/<name>/librmhip.so (git-ignored; travels to the GPU box).  Select it with RMHIP_LIBRARY=ab_old/<name>/librmhip.so.
# Usage: scripts/build_variant.sh <name> "<flags>" [unit.hip]
set -e
NAME=$1; FLAGS=$2; UNIT=${3:-lu.hip}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/runmat_amd/csrc
OUT=$ROOT/ab_old/$NAME
mkdir -p "$OUT"
OBJ=$OUT/${UNIT%.*}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Wno-unused-function -Wno-unused-result -I$ROOT/include $FLAGS -c $SRC/$UNIT -o $OBJ
OTHERS=$(ls $SRC/*.o | grep -v "/${UNIT%.*}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 $OTHERS $OBJ -shared -L/opt/rocm/lib -lhiprtc -ldl -lrt -lpthread -Wl,-rpath,/opt/rocm/lib -o $OUT/librmhip.so
echo "built $OUT/librmhip.so"
