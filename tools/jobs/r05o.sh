#!/usr/bin/env bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05o
for lib in libl2q.so libl2q_ncg2.so; do
L2Q_LIB_NAME=$lib timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "kstream" 2>&1 | tail -2
L2Q_LIB_NAME=$lib timeout 300 python tools/time_heads_h.py 2 2>&1 | grep "^\["
done
