#!/usr/bin/env bash
# round 5, second pass: mixed-precision training tests, the published U(1) configuration, 2-rank gloo bench
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r05b; mkdir -p $o
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_trainer_gpu.py -q -m gpu -x > $o/t_train.log 2>&1; echo "train rc=$?" | tee -a $o/summary.txt
tail -5 $o/t_train.log
python - > $o/published.json 2> $o/published.err <<'PY'
import sys, json
sys.path.insert(0, 'l2hmc-qcd_amd'); sys.path.insert(0, '.')
import torch
import bench
print(json.dumps(bench.published_u1(), indent=1))
PY
echo "published rc=$?" | tee -a $o/summary.txt
cat $o/published.json; tail -5 $o/published.err
L2Q_BENCH_SHARE_GPU=1 L2Q_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --nchains 64 > $o/bench_2ranks_gloo.json 2> $o/bench_2ranks.err; echo "2rank rc=$?" | tee -a $o/summary.txt
python - $o <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + '/bench_2ranks_gloo.json').readline())
print(d['value'], d['n_gpus'], d['rccl_ranks'], d.get('per_rank'), d.get('grad_allreduce_probe'))
PY
tail -3 $o/bench_2ranks.err
