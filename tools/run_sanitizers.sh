#!/bin/bash
# CPU-side sanitizer pass: AddressSanitizer + UndefinedBehaviorSanitizer over everything of the path that
# runs on the host — the oracle (g++), the host code of libisonclust2_hip.so (driver, host aligner, consensus driver,
# C ABI argument / error paths; clang's host ASan, the device code is compiled as usual), and the command line with its
# .cer reader / writer (round trip, every truncation, flipped bytes, crafted counts: `isONclust2-hip selftest`).
# GPU ASan is not available on this pool: the kernels themselves are covered by the parity tests on the MI355X.
# Two passes, because the two ASan runtimes (gcc's for the oracle, clang's for the HIP library) cannot share a process.
#   tools/run_sanitizers.sh            -> log in profiles/r04_sanitizers.log
# Exit 0 only when every build and every pass ran to its end AND nothing was reported: the braces below run in a pipeline
# subshell, so their `exit 1` is read back through PIPESTATUS, the '== done' marker is required, and a pytest summary with
# anything but passes (failed, error) counts as a finding.
set -u -o pipefail
cd "$(dirname "$0")/.."
OUT=build/asan
mkdir -p "$OUT"
LOG=profiles/${SAN_LOG:-r04_sanitizers.log}
CLANG_RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
GCC_RT=$(gcc -print-file-name=libasan.so)
# (libstdc++ next to the runtime: preloaded into python, ASan's __cxa_throw interceptor otherwise finds no real one)
STDCXX=$(gcc -print-file-name=libstdc++.so.6)
SRC=isonclust2_amd/csrc
{
echo "== sanitizer pass $(date -u +%Y-%m-%dT%H:%MZ) =="
echo "-- build: oracle with g++ -fsanitize=address,undefined"
g++ -std=c++14 -O1 -g -DNDEBUG -msse3 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $OUT/liboracle_asan.so oracle/oracle.cpp oracle/poa_oracle.cpp oracle/sg_striped.cpp || exit 1
echo "-- build: libisonclust2_hip.so host code with clang -fsanitize=address,undefined (device code unchanged)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -pthread -ffp-contract=off -Wno-unused-result -Wno-option-ignored \
    -fsanitize=address,undefined -fno-omit-frame-pointer -shared-libsan -Iinclude -I$SRC -shared -o $OUT/libisonclust2_hip.so \
    $SRC/ioc_kernels.hip $SRC/ioc_score.hip $SRC/ioc_resolve.hip $SRC/ioc_extract.hip $SRC/ioc_capi.cpp $SRC/ioc_host.cpp $SRC/ioc_align.cpp $SRC/ioc_align_gpu.hip \
    $SRC/ioc_update.hip $SRC/ioc_consensus.cpp $SRC/ioc_poa.hip $SRC/ioc_sort.hip $SRC/ioc_sort_long.hip $SRC/ioc_dist.cpp $SRC/ioc_build_sort.hip -L/opt/rocm/lib -lrccl || exit 1
echo "-- build: command line (main.cpp, cer.cpp) with the same runtime"
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared-libsan -Iinclude \
    -o $OUT/isONclust2-hip $SRC/cli/main.cpp $SRC/cli/cer.cpp -L$OUT -lisonclust2_hip -L/opt/rocm/lib -lamdhip64 \
    -Wl,-rpath,"$PWD/$OUT" -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,"$(dirname "$CLANG_RT")" || exit 1
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "-- pass A: pytest -m 'not gpu' with the ORACLE under ASan/UBSan"
LD_PRELOAD="$GCC_RT $STDCXX" ORACLE_LIB=$PWD/$OUT/liboracle_asan.so python -m pytest tests -q -m "not gpu" -p no:cacheprovider > $OUT/passA.log 2>&1
grep -E "runtime error:|ERROR: AddressSanitizer|CHECK failed" $OUT/passA.log | head -5; tail -2 $OUT/passA.log
tail -1 $OUT/passA.log | grep -Eq "^[0-9]+ passed(, [0-9]+ (deselected|skipped|warnings?))* in " || { echo "PASS A DID NOT END IN A PASSED-ONLY SUMMARY"; exit 1; }
echo "-- pass B: pytest -m 'not gpu' with the HIP LIBRARY's host code and the command line under ASan/UBSan"
LD_PRELOAD="$CLANG_RT $STDCXX" IOC_LIB=$PWD/$OUT/libisonclust2_hip.so IOC_CLI=$PWD/$OUT/isONclust2-hip python -m pytest tests -q -m "not gpu" -p no:cacheprovider > $OUT/passB.log 2>&1
grep -E "runtime error:|ERROR: AddressSanitizer|CHECK failed" $OUT/passB.log | head -5; tail -2 $OUT/passB.log
tail -1 $OUT/passB.log | grep -Eq "^[0-9]+ passed(, [0-9]+ (deselected|skipped|warnings?))* in " || { echo "PASS B DID NOT END IN A PASSED-ONLY SUMMARY"; exit 1; }
echo "-- pass C: .cer round trip, every truncation, flipped bytes, crafted counts"
LD_PRELOAD= $OUT/isONclust2-hip selftest /tmp/ioc_asan_selftest.cer > $OUT/passC.log 2>&1 || { tail -5 $OUT/passC.log; echo "PASS C FAILED"; exit 1; }
tail -3 $OUT/passC.log
echo "== done: no sanitizer report above means clean =="
} 2>&1 | tee "$LOG"
RC=${PIPESTATUS[0]}
[ "$RC" -eq 0 ] || { echo "SANITIZER PASS BROKE OFF (exit $RC)"; exit 1; }
grep -q "^== done" "$LOG" || { echo "SANITIZER PASS DID NOT REACH ITS END"; exit 1; }
grep -q "ERROR: AddressSanitizer\|runtime error:\|CHECK failed\|[0-9] failed\|[0-9] error" "$LOG" && { echo "SANITIZER FINDINGS"; exit 1; }
exit 0
