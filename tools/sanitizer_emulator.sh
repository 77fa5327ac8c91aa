#!/bin/bash
# Build the CPU emulator of the kernel sources under a sanitizer and run emulator tests against it (no GPU).
#   tools/sanitizer_emulator.sh address|undefined|thread [pytest args ...]
# address / undefined: device buffers are exactly-sized heap blocks in the emulator, so a kernel's out-of-bounds access or undefined arithmetic is a report with a
# source line (profiles/r05_asan_emulator.txt).  thread: for use as SVT_HIP_LIB of a ThreadSanitizer build of the reference encoder
# (make -C oracle enc OUT=/tmp/tsan CC="gcc -fsanitize=thread -g"; profiles/r05_tsan_seams.txt) -- tests/emu/hipemu.h announces its lane fibers to TSan in such builds.
set -e
SAN=${1:?address|undefined|thread}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${SAN_OUT:-/tmp/svt_hip_emu_$SAN}
mkdir -p "$OUT"
FLAGS="-O1 -g -fno-omit-frame-pointer -fsanitize=$SAN"
[ "$SAN" = address ] && FLAGS="$FLAGS -fsanitize-recover=address"
[ "$SAN" = undefined ] && FLAGS="$FLAGS -fno-sanitize=vptr"
cd "$ROOT/svt-av1-psy_amd/csrc"
ls *.hip | xargs -P "$(nproc)" -I{} g++ $FLAGS -std=c++17 -fPIC -ffp-contract=off -Wno-unknown-pragmas -Wno-attributes -I../../tests/emu -I../../include -x c++ -c {} -o "$OUT/{}.o"
g++ -shared -fPIC -fsanitize=$SAN -o "$OUT/libsvtav1_hipemu.so" "$OUT"/*.o
echo "built $OUT/libsvtav1_hipemu.so"
[ "$SAN" = thread ] && exit 0
cd "$ROOT"
RT=$(gcc -print-file-name=lib$([ "$SAN" = address ] && echo asan || echo ubsan).so)
export ASAN_OPTIONS="detect_leaks=0:halt_on_error=0:detect_stack_use_after_return=0:log_path=$OUT/report" UBSAN_OPTIONS="print_stacktrace=0:log_path=$OUT/report"
rm -f "$OUT"/report.*
SVT_HIP_EMU_LIB="$OUT/libsvtav1_hipemu.so" LD_PRELOAD="$RT" python -m pytest -q -m "not gpu" -p no:cacheprovider "${@:-tests/test_cdef.py}" || true
echo "reports:"; cat "$OUT"/report.* 2>/dev/null | grep -E "SUMMARY|runtime error" | sort | uniq -c | sort -rn | head -40
