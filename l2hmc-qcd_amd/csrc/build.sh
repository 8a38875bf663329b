#!/usr/bin/env bash
# Build libl2q.so (all HIP kernels + the C ABI) for gfx950, in-tree.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../l2hmc/_lib"
mkdir -p "$out" "$here/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
pids=()
newest_hdr="$(ls -t "$here"/*.hpp "$here/../../include/l2q.h" | head -1)"
for f in common su3_kernels su3_force_rows su3_force_nu u1_kernels u1_fused gemm gemm_f16 train_kernels su3_train_kernels su3_rect_kernels; do
  # rebuild when the source or ANY header is newer than the object
  if [ ! -f "$here/obj/$f.o" ] || [ "$here/$f.hip" -nt "$here/obj/$f.o" ] \
     || [ "$newest_hdr" -nt "$here/obj/$f.o" ]; then
    $HIPCC $FLAGS -c "$here/$f.hip" -o "$here/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out/libl2q.so" "$here"/obj/{common,su3_kernels,su3_force_rows,su3_force_nu,u1_kernels,u1_fused,gemm,gemm_f16,train_kernels,su3_train_kernels,su3_rect_kernels}.o
echo "built $out/libl2q.so"
