#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/e
o=gpurun_out/e
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $o/gpu_tests.log 2>&1; echo "rc=$?" >> $o/gpu_tests.log
timeout 600 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?" >> $o/bench.err
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "rc=$?" >> $o/smoke.log
tail -25 $o/gpu_tests.log; head -c 300 $o/bench.json; tail -3 $o/smoke.log
