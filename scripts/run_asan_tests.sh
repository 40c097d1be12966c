#!/bin/bash
# The CPU test-suite's library-facing tests (front end: WGSL parse / code generation / hipRTC compile checks, the ABI table, the C++ host
# mirror's no-device paths) against the AddressSanitizer build of librmhip.so (scripts/build_asan.sh).  Leak checking is off: the
# interpreter and hipRTC keep allocations for the life of the process.  Exit status = pytest's; an ASAN report aborts the process.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$ROOT/ab_old/asan/librmhip.so" ] || { echo "build first: scripts/build_asan.sh"; exit 2; }
cd "$ROOT"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 RMHIP_LIBRARY=$ROOT/ab_old/asan/librmhip.so \
  python -m pytest tests/test_front_end.py tests/test_bindings.py tests/test_auto_offload.py -x -q -m "not gpu" "$@"
