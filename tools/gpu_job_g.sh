#!/usr/bin/env bash
# round-2 (second session) measurement job: GPU tests, bench (eval / hmc / train), rocprof kernel
# stats of the bench command, PMC passes on the shipping kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/g
o=gpurun_out/g
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; tail -2 $o/pytest_gpu.log
timeout 600 python bench.py > $o/bench.json 2> $o/bench.err; echo "rc=$?" >> $o/bench.err
timeout 600 python bench.py --mode hmc --no-cpu-baseline > $o/bench_hmc.json 2>> $o/bench.err
timeout 900 python bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > $o/bench_train.json 2>> $o/bench.err
bash tools/kstats.sh $o/bench_l2hmc_kernel_stats.txt python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-spot-check > /dev/null
bash tools/kstats.sh $o/train_su3_kernel_stats.txt python $GRAFT_REPO_ROOT/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-spot-check > /dev/null
bash tools/pmc_collect.sh r02j > $o/pmc.log 2>&1
cp profiles/r02j_pmc_counters.txt profiles/pmc_traffic.json $o/ 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-spot-check > $o/bench_after_pmc.json 2>> $o/bench.err
head -c 600 $o/bench.json; echo; tail -3 $o/pmc.log
bash tools/bench_configs.sh > $o/u1_configs.txt 2>&1; tail -4 $o/u1_configs.txt
