#!/usr/bin/env bash
# round-2 GPU job A: size-parity tests, bench (eval / train / self-spawn), fresh PMC passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/a
o=gpurun_out/a
{ free -g | head -2; nproc; rocm-smi --showmeminfo vram | head -8; } > $o/box.txt 2>&1
timeout 1500 python -m pytest tests/test_sizes_gpu.py -x -q -k "not l2hmc_trajectory_16" --durations=10 > $o/sizes.log 2>&1; echo "sizes rc=$?" >> $o/sizes.log
timeout 600 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?" >> $o/bench.err
L2Q_BENCH_SHARE_GPU=1 L2Q_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 > $o/bench_spawn2.json 2> $o/bench_spawn2.err; echo "rc=$?" >> $o/bench_spawn2.err
timeout 600 python bench.py --gpus 2 > $o/bench_gpus2_refused.txt 2>&1; echo "rc=$?" >> $o/bench_gpus2_refused.txt
timeout 900 python bench.py --mode train --steps 3 --warmup 1 > $o/bench_train.json 2> $o/bench_train.err; echo "rc=$?" >> $o/bench_train.err
L2Q_BENCH_SHARE_GPU=1 L2Q_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --nchains 64 --mode train --steps 2 --warmup 1 > $o/bench_train2.json 2> $o/bench_train2.err; echo "rc=$?" >> $o/bench_train2.err
bash tools/pmc_collect.sh r02a > $o/pmc.log 2>&1
cp profiles/r02a_pmc_counters.txt profiles/pmc_traffic.json $o/ 2>/dev/null
tail -3 $o/sizes.log; cat $o/bench.json | head -c 600; tail -2 $o/bench_train.err
