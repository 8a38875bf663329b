#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04h; mkdir -p $o
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_sizes_gpu.py -q -m gpu -k "heads_sliced or cfg4 or cfg5_l2hmc" > $o/t_heads.log 2>&1; echo "heads tests rc=$?" | tee -a $o/summary.txt
for rep in 1 2 3; do
  L2Q_LIB_NAME=libl2q_hsold.so BOTH=0 python tools/time_heads_sliced.py 2>&1 | tail -1 | tee -a $o/heads_ab.txt
  BOTH=0 python tools/time_heads_sliced.py 2>&1 | tail -1 | tee -a $o/heads_ab.txt
done
tail -3 $o/t_heads.log
