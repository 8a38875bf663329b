#!/usr/bin/env bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/lds
for rep in 1 2; do for n in ldsbase ldsvol; do
  echo "== $n"; L2Q_LIB_NAME=libl2q_$n.so python tools/time_gemm_in.py 2>&1 | grep -v amdgpu | tail -1
  L2Q_LIB_NAME=libl2q_$n.so python tools/bench_heads_f64.py 2>&1 | grep -v amdgpu | tail -3
done; done | tee gpurun_out/lds/out.txt
L2Q_LIB_NAME=libl2q_ldsvol.so timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm or heads" -x 2>&1 | tail -3
