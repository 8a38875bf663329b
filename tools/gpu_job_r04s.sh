#!/usr/bin/env bash
# round-4 session 3: training tape on the TAPE instances of the sliced heads kernel
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04s; mkdir -p $o
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_trainer_gpu.py tests/test_abi.py -q -m gpu -x > $o/t.log 2>&1; echo "tests rc=$?" | tee -a $o/summary.txt
tail -5 $o/t.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "sliced or heads" > $o/t2.log 2>&1; echo "sliced tests rc=$?" | tee -a $o/summary.txt
tail -3 $o/t2.log
for flag in "" "--fp64-train-heads" "" "--fp64-train-heads"; do
  f=$o/bench_train${flag:+_fp64heads}.json
  python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe $flag > $f 2> $o/bench_train.err; echo "train '$flag' rc=$?" | tee -a $o/summary.txt
  python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print(sys.argv[1], d['value'], d['ms_per_step'])
for k, v in list(d['kernels'].items())[:12]:
    print('   ', k, v)
PY
done
