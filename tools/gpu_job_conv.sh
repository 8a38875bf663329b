#!/usr/bin/env bash
# half-precision conv stack: parity tests, then the per-kernel time of the cfg-3 default-network trajectory
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/kt
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_dynamics_gpu.py -m gpu -x -q -k "conv or bf16 or half" 2>&1 | tail -3
bash tools/kstats.sh gpurun_out/kt/conv_patch.txt python $GRAFT_REPO_ROOT/tools/bench_u1.py --L 64 64 --nb 8192 --beta 6 --steps 1 --no-hmc --no-graph --conv --precision fp16 "$@" 2>&1 | head -10 | cut -c1-150
grep "Dynamics.forward" /tmp/kstats.*/stdout.log | tail -1
