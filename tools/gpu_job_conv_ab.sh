#!/usr/bin/env bash
# per-kernel times of the cfg-3 conv trajectory for alternate builds: tools/gpu_job_conv_ab.sh <lib names...>
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/kt
for n in "$@"; do
  echo "== $n"
  L2Q_LIB_NAME=libl2q_$n.so bash tools/kstats.sh gpurun_out/kt/conv_$n.txt python $GRAFT_REPO_ROOT/tools/bench_u1.py --L 64 64 --nb 8192 --beta 6 --steps 1 --no-hmc --no-graph --conv --precision fp16 2>&1 | grep "conv_patch\|conv_gemm" | cut -c1-20,40-75,100-150
done
