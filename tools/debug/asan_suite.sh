#!/bin/bash
# Host side of libfigdraw_hip.so under AddressSanitizer (CPU suite on record-only contexts; GPU sanitizers are not available on
# the pool).  usage: tools/debug/asan_suite.sh [pytest args]     -> build/libfigdraw_hip_asan.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT/figdraw_amd/csrc"
make -s -j8 >/dev/null   # the device objects come from the product build
mkdir -p "$ROOT/build/obj/asan"
for f in fdh_context fdh_record fdh_frontend fdh_walkpool fdh_capi fdh_comm; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fvisibility=hidden -Xarch_host -msse4.1 -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
    -DFDH_SPLIT_UNIFORM=1 -x hip -c $f.cpp -o "$ROOT/build/obj/asan/$f.o" 2> >(grep -v "recognized feature" >&2) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -fsanitize=address "$ROOT"/build/obj/product/k_*.o "$ROOT"/build/obj/asan/*.o -ldl -lpthread -o "$ROOT/build/libfigdraw_hip_asan.so"
cd "$ROOT"
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0 FIGDRAW_HIP_LIB="$ROOT/build/libfigdraw_hip_asan.so" python -m pytest tests -m "not gpu" -q --deselect tests/test_abi_and_sharding.py "$@"
