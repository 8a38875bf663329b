#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/d
o=gpurun_out/d
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "stencils or cold_start or ops_golden" > $o/stencil.log 2>&1; echo "rc=$?" >> $o/stencil.log
timeout 600 python tools/force_bench.py > $o/force_bench.log 2>&1; echo "rc=$?" >> $o/force_bench.log
tail -4 $o/stencil.log; cat $o/force_bench.log
