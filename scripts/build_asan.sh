#!/bin/bash
# AddressSanitizer build of the HOST side of librmhip.so (device code is compiled as usual: GPU ASAN needs xnack+, which this pool
# does not offer) into ab_old/asan/librmhip.so, for the CPU test-suite:
#   scripts/build_asan.sh && scripts/run_asan_tests.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/ab_old/asan
rm -rf "$OUT" && mkdir -p "$OUT/tree/runmat_amd" "$OUT/tree/include"
cp -r "$ROOT/runmat_amd/csrc" "$OUT/tree/runmat_amd/csrc"
cp "$ROOT"/include/*.h "$ROOT"/include/*.hpp "$OUT/tree/include/" 2>/dev/null || true
cd "$OUT/tree/runmat_amd/csrc"
rm -f *.o *.so *_str.inc
make -j8 CXXFLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Wno-unused-function -Wno-unused-result -I../../include -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer" \
        LDFLAGS="-shared -fsanitize=address -shared-libasan -L/opt/rocm/lib -lhiprtc -ldl -lrt -lpthread -Wl,-rpath,/opt/rocm/lib" > "$OUT/build.log" 2>&1 || { tail -20 "$OUT/build.log"; exit 1; }
cp librmhip.so "$OUT/librmhip.so"
cd "$ROOT" && rm -rf "$OUT/tree"
echo "built $OUT/librmhip.so"
