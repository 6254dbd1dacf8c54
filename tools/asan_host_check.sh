#!/usr/bin/env bash
# Host-side AddressSanitizer build of libvlg (sanitizers run on the CPU build only - GPU ASan / xnack+ code objects
# are not available on this pool).  Device code is compiled without instrumentation (-fno-gpu-sanitize); the host code of every
# translation unit (handle management, weight store, launch logic, argument validation) is instrumented.  Runs the CPU-side C-ABI tests
# against the instrumented library with the ASan runtime preloaded into the (uninstrumented) Python interpreter.
#   tools/asan_host_check.sh            -> build/asan/libvlg.so, then pytest tests/test_cabi_cpu.py tests/test_io_cpu.py tests/test_serve_cpu.py
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/build/asan"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
pids=()
for src in "$ROOT"/video-llamagen_amd/csrc/*.hip; do
  obj="$OUT/$(basename "${src%.hip}").o"
  "$HIPCC" --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -shared-libsan -c "$src" -o "$obj" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan -o "$OUT/libvlg.so" "$OUT"/*.o
echo "built $OUT/libvlg.so"
cd "$ROOT"
# detect_leaks=0: CPython itself "leaks" at exit; the checks of interest are out-of-bounds / use-after-free in the library's host code
VLG_LIB_PATH="$OUT/libvlg.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:protect_shadow_gap=0 \
  python -m pytest tests/test_cabi_cpu.py tests/test_io_cpu.py tests/test_serve_cpu.py -x -q -p no:cacheprovider -m "not gpu"
