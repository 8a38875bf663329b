#!/usr/bin/env bash
# round 4, second pass: pair force kernel (parity of every variant + A/B timing at 8^4 and 16^4), the
# from-seed trajectories, the reference script's call sequence, the half-precision fused-heads diagnostic
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04b; mkdir -p $o
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "stencils" > $o/t_stencil.log 2>&1; echo "stencils rc=$?" | tee -a $o/summary.txt
timeout 600 python tools/force_bench.py --big > $o/force_ab.txt 2>&1; echo "force_bench rc=$?" | tee -a $o/summary.txt
timeout 900 python -m pytest tests/test_dynamics_gpu.py tests/test_trainer_gpu.py -q -m gpu -k "from_seed or train4dsu3" > $o/t_seed.log 2>&1; echo "from_seed rc=$?" | tee -a $o/summary.txt
timeout 300 python tools/dbg_half_fused.py bf16 > $o/dbg_half.txt 2>&1; echo "dbg rc=$?" | tee -a $o/summary.txt
timeout 1500 python -m pytest tests -q -m gpu > $o/t_all.log 2>&1; echo "all rc=$?" | tee -a $o/summary.txt
tail -3 $o/t_stencil.log; grep -v amdgpu.ids $o/force_ab.txt | grep "force_tile=[65]"; tail -3 $o/t_seed.log; cat $o/dbg_half.txt | tail -12; tail -8 $o/t_all.log
