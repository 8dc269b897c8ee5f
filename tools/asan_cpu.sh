#!/bin/bash
# Host code of libucg_hip.so (table / settings parsers, table build, the I/O formats, the C ABI's host side) under
# AddressSanitizer + UBSan, on the CPU tier of the tests.  Device code is not instrumented (-fno-gpu-sanitize): GPU
# sanitizers are not available on this pool.  Objects and the library go to /tmp/ucg_asan, nothing into the tree.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${UCG_ASAN_DIR:-/tmp/ucg_asan}
mkdir -p "$OUT"
cd "$ROOT/lammps-ucg-dev_amd/csrc"
F="-O1 -g -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -Wno-unused-result"
for f in ucg_model.cpp ucg_io.cpp ucg_pair.hip ucg_pair_hot.hip ucg_pair_vrow.hip ucg_density.hip ucg_fix.hip ucg_ranmars.hip ucg_neigh.hip ucg_cluster.hip ucg_capi.hip ucg_comm.hip ucg_host.hip; do
  /opt/rocm/bin/hipcc $F -c "$f" -o "$OUT/${f%.*}.o" &
done
/opt/rocm/bin/hipcc $F -ffp-contract=fast -DUCG_FUSED -c ucg_pair.hip -o "$OUT/ucg_pair_fused.o" &
/opt/rocm/bin/hipcc $F -ffp-contract=fast -DUCG_FUSED -c ucg_pair_hot.hip -o "$OUT/ucg_pair_hot_fused.o" &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan \
  -o "$OUT/libucg_hip_asan.so" "$OUT"/*.o -ldl
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$ROOT"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  UCG_HIP_LIBRARY="$OUT/libucg_hip_asan.so" python -m pytest tests -x -q -m "not gpu" "$@"
