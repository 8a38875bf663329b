#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04u; mkdir -p $o
export TMPDIR=/tmp
python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "sliced_heads" 2>&1 | grep -E "^E|passed|failed" | head -20
for kt in 0 1; do
for flag in "" "--fp64-train-heads"; do
  f=$o/bench_train_kt${kt}${flag:+_fp64heads}.json
  L2Q_BENCH_NO_KTIMER=$kt python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe $flag > $f 2> $o/bench_train.err
  python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('kernel_time_fraction_of_wall'))
PY
done; done
