#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04w; mkdir -p $o
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_trainer_gpu.py tests/test_abi.py -q -m gpu -x > $o/t.log 2>&1; echo "train tests rc=$?" | tee -a $o/summary.txt
tail -3 $o/t.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "sliced or heads" > $o/t2.log 2>&1; echo "sliced tests rc=$?" | tee -a $o/summary.txt
tail -3 $o/t2.log
for i in 1 2; do
python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe > $o/bench_train_$i.json 2> $o/bench_train.err; echo "train rc=$?" | tee -a $o/summary.txt
python - $o/bench_train_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('instrumented_ms_per_step'))
for k, v in list(d['kernels'].items())[:14]:
    print('   ', k, v)
PY
done
