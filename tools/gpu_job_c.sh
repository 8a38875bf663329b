#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/c
o=gpurun_out/c
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "stencils or cold_start or ops_golden" > $o/stencil.log 2>&1; echo "rc=$?" >> $o/stencil.log
timeout 600 python tools/force_bench.py --big > $o/force_bench.log 2>&1; echo "rc=$?" >> $o/force_bench.log
timeout 900 python tools/cal_cfg5.py hot > $o/cal_hot.log 2>&1; echo "rc=$?" >> $o/cal_hot.log
timeout 600 python tools/cal_cfg5.py warm > $o/cal_warm.log 2>&1; echo "rc=$?" >> $o/cal_warm.log
timeout 600 python bench.py --no-cpu-baseline > $o/bench.json 2> $o/bench.err; echo "bench rc=$?" >> $o/bench.err
tail -4 $o/stencil.log; cat $o/force_bench.log; grep -v "^$" $o/cal_hot.log | tail -12; tail -4 $o/cal_warm.log
