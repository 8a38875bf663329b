#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/b
o=gpurun_out/b
timeout 900 python tools/cal_cfg5.py > $o/cal_cfg5.log 2>&1; echo "rc=$?" >> $o/cal_cfg5.log
timeout 2400 python -m pytest tests -m gpu -q --durations=15 -k "not l2hmc_trajectory_16" > $o/gpu_tests.log 2>&1; echo "rc=$?" >> $o/gpu_tests.log
timeout 600 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?" >> $o/bench.err
tail -5 $o/cal_cfg5.log; tail -5 $o/gpu_tests.log; head -c 400 $o/bench.json
