#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r05e; mkdir -p $o
export TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
for n in prof1 prof2; do
  echo "== $n"; BOTH=0 L2Q_LIB_NAME=libl2q_$n.so timeout 300 python3 tools/time_heads_sliced.py 2>&1 | grep -E "sliced prof|pair sliced" | tail -3
done 2>&1 | tee $o/prof.txt
timeout 600 python -m pytest tests/test_dynamics_gpu.py -q -m gpu -x -k "auto_graphed" 2>&1 | tail -3
