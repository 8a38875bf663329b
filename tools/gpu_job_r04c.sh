#!/usr/bin/env bash
# force kernel with row pipelining: parity of every variant, A/B timing
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04c; mkdir -p $o
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "stencils or heads_sliced" > $o/t_stencil.log 2>&1; echo "stencils rc=$?" | tee -a $o/summary.txt
timeout 600 python tools/force_bench.py --big > $o/force_ab.txt 2>&1; echo "force_bench rc=$?" | tee -a $o/summary.txt
tail -3 $o/t_stencil.log; grep -v amdgpu.ids $o/force_ab.txt | grep "force_tile=[654]"
