#!/bin/sh
# AddressSanitizer and ThreadSanitizer over the HOST code of libgtsam_amd.so (symbolic analysis on host threads, schedule
# construction, launch issue, file formats) without a GPU: the host side of every .hip file is instrumented
# (-Xarch_host -fsanitize=...), the device side is compiled as usual, and the library runs under tools/hipstub.
#   sh tools/sanitize/host_sanitizers.sh [workloads...]      (default: a small, a sharded-relevant and the headline shape)
# Round 1: both clean on edge, dubrovnik_3_7, bal:60:6000:7, bal:300:20000:3, sphere2500, ladybug1723, w20000
# (GTG_HOST_THREADS=8; the default set below also with GTG_ND_DEPTH=2).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)")
WORK=${WORK:-edge bal:60:6000:7 bal:300:20000:3 sphere2500 ladybug1723}
[ $# -gt 0 ] && WORK="$*"
gcc -O2 -fPIC -shared -o "$ROOT/tools/hipstub/libhipstub.so" "$ROOT/tools/hipstub/hipstub.c"
for SAN in address thread; do
  OUT=/tmp/gtsam_amd_$SAN; mkdir -p $OUT
  for f in api cholesky assemble factors pcg; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Xarch_host -fsanitize=$SAN -Xarch_host -fno-omit-frame-pointer \
        -c "$ROOT/gtsam_amd/csrc/$f.hip" -o $OUT/$f.o 2>/dev/null
  done
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -fsanitize=$SAN -c "$ROOT/gtsam_amd/csrc/io.cpp" -o $OUT/io.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Xarch_host -fsanitize=$SAN -o $OUT/libgtsam_amd.so $OUT/*.o 2>/dev/null
  if [ $SAN = address ]; then LIBRT=$RT/libclang_rt.asan-x86_64.so; else LIBRT=$RT/libclang_rt.tsan-x86_64.so; fi
  for ND in "" 2; do
    echo "== $SAN sanitizer, GTG_ND_DEPTH='$ND'"
    env SAN_LIB=$OUT/libgtsam_amd.so GTG_HOST_THREADS=8 ${ND:+GTG_ND_DEPTH=$ND} ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
      TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0" LD_PRELOAD="$LIBRT $ROOT/tools/hipstub/libhipstub.so" \
      python "$ROOT/tools/sanitize/run_host_paths.py" $WORK
  done
done
echo "host sanitizers: clean"
