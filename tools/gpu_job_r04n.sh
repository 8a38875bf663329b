#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04n; mkdir -p $o
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm_sliced" -x > $o/t_gs.log 2>&1; echo "gemm_sliced tests rc=$?" | tee -a $o/summary.txt
timeout 300 python tools/time_gemm_sliced.py > $o/time_gs.txt 2>&1
tail -4 $o/t_gs.log; grep -v amdgpu $o/time_gs.txt
