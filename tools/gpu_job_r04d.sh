#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04d; mkdir -p $o
export TMPDIR=/tmp
timeout 600 python tools/force_stagger.py > $o/stagger.txt 2>&1; echo "stagger rc=$?" | tee -a $o/summary.txt
timeout 900 python -m pytest tests/test_dynamics_gpu.py tests/test_kernels_gpu.py -q -m gpu -k "half_precision_networks or save_load or heads_sliced" > $o/t_fix.log 2>&1; echo "fix rc=$?" | tee -a $o/summary.txt
grep -v amdgpu.ids $o/stagger.txt; tail -3 $o/t_fix.log
