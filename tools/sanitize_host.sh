#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of libgaudi_hip.so (CPU box only: GPU ASan / XNACK runs
# are not available on the pool).  The library is rebuilt with the sanitizers on the host compilation only
# (-Xarch_host -fsanitize=address,undefined; the device code objects are the normal ones), and the device-free parts of the
# CPU suite -- the ABI / packer tests (incl. the concurrent-packing test), the host logic tests, the 8-wave graph metadata
# tests -- run against it through GAUDI_LIB with the ASan runtime preloaded into the (uninstrumented) python process.
#   tools/sanitize_host.sh > profiles/r03_host_sanitizers.txt 2>&1
set -e
cd "$(dirname "$0")/.."
CLANG_LIB=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
ASAN_RT=$CLANG_LIB/libclang_rt.asan-x86_64.so
OUT=$PWD/gaudi_amd/libgaudi_hip_asan.so
OBJ=/tmp/gaudi_asan_obj
mkdir -p $OBJ
SAN="-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -shared-libsan"
echo "== build (sanitizers on the host compilation: $SAN)"
cd gaudi_amd/csrc
pids=""
for tu in gaudi_hip kern_edm_192 kern_fused_192_208 kern8_edm_192 kern8_fused_192_208 kern8s_edm_192 kern8s_fused_192_208 kern8h_fused_192_208; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -g -std=c++17 -fPIC -fno-slp-vectorize $SAN -DGAUDI_STAMP_STUBS -w -c $tu.hip -o $OBJ/$tu.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $OUT $OBJ/*.o
cd ../..
ls -la $OUT
nm -D $OUT | grep -c "__asan_\|__ubsan_" | sed 's/^/sanitizer runtime symbols referenced: /'
echo "== device-free CPU tests against the sanitised library"
GAUDI_LIB=$OUT LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python3 -m pytest tests/test_abi_cpu.py tests/test_host_logic.py tests/test_meta8_cpu.py -q -p no:cacheprovider 2>&1 | tail -15
echo "== done"
