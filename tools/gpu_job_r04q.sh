#!/usr/bin/env bash
# round-4 session 3: Cayley-Hamilton Frechet derivative + Jacobi early exit in the SU(3) training kernels
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04q; mkdir -p $o
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_trainer_gpu.py -q -m gpu -x > $o/t_train.log 2>&1; echo "train tests rc=$?" | tee -a $o/summary.txt
tail -3 $o/t_train.log
for lib in libl2q.so libl2q_series.so libl2q.so libl2q_series.so; do
  L2Q_LIB_NAME=$lib python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe > $o/bench_train_$lib.json 2> $o/bench_train.err; echo "train $lib rc=$?" | tee -a $o/summary.txt
  python - $o/bench_train_$lib.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print(sys.argv[1], d['value'], d['ms_per_step'])
for k, v in list(d['kernels'].items())[:9]:
    print('   ', k, v)
PY
done
