#!/usr/bin/env bash
# round-4 session 3: expm bwd LDS stash A/B, force epilogue (diag real parts not formed) A/B + tests
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04r; mkdir -p $o
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py tests/test_sizes_gpu.py -q -m gpu -x > $o/t.log 2>&1; echo "tests rc=$?" | tee -a $o/summary.txt
tail -3 $o/t.log
for i in 1 2 3; do
  for lib in libl2q.so libl2q_fulldiag.so; do L2Q_LIB_NAME=$lib python tools/force_time5.py 2>&1 | tee -a $o/force_ab.txt; done
done
for lib in libl2q.so libl2q_nostash.so libl2q.so libl2q_nostash.so; do
  L2Q_LIB_NAME=$lib python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe > $o/bench_train_$lib.json 2> $o/bench_train.err; echo "train $lib rc=$?" | tee -a $o/summary.txt
  python - $o/bench_train_$lib.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print(sys.argv[1], d['value'], d['ms_per_step'])
for k, v in list(d['kernels'].items())[:6]:
    print('   ', k, v)
PY
done
