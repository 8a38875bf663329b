#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04ae; mkdir -p $o
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_trainer_gpu.py -q -m gpu -x > $o/t.log 2>&1; echo "train tests rc=$?" | tee -a $o/summary.txt
grep -E "^E " $o/t.log | head -8 | cut -c1-200; tail -3 $o/t.log
for flag in "" "--separate-v-pairs" "" "--separate-v-pairs"; do
f=$o/bench_train${flag:+_separate}.json
python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe $flag > $f 2> $o/bench_train.err; echo "train '$flag' rc=$?" | tee -a $o/summary.txt
python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('instrumented_ms_per_step'))
for k in ('l2q_v_update_bwd_pair_c128', 'l2q_v_update_bwd_acc_c128', 'l2q_v_update_bwd_c128'):
    if k in d['kernels']: print('   ', k, d['kernels'][k])
PY
done
tail -3 $o/bench_train.err | cut -c1-200
