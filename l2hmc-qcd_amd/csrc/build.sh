#!/usr/bin/env bash
# Build libl2q.so (all HIP kernels + the C ABI) for gfx950, in-tree.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../l2hmc/_lib"
mkdir -p "$out" "$here/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
pids=()
for f in common su3_kernels u1_kernels gemm train_kernels su3_train_kernels; do
  if [ ! -f "$here/obj/$f.o" ] || [ "$here/$f.hip" -nt "$here/obj/$f.o" ] \
     || [ "$here/l2q_common.hpp" -nt "$here/obj/$f.o" ] || [ "$here/su3_math.hpp" -nt "$here/obj/$f.o" ] || [ "$here/u1_math.hpp" -nt "$here/obj/$f.o" ] || [ "$here/su3_links.hpp" -nt "$here/obj/$f.o" ] \
     || [ "$here/../../include/l2q.h" -nt "$here/obj/$f.o" ]; then
    $HIPCC $FLAGS -c "$here/$f.hip" -o "$here/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out/libl2q.so" "$here"/obj/{common,su3_kernels,u1_kernels,gemm,train_kernels,su3_train_kernels}.o
echo "built $out/libl2q.so"
