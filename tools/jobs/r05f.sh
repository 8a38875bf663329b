#!/usr/bin/env bash
# round 5 checkpoint: whole GPU tier, the default bench line (all blocks), train bench
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r05f; mkdir -p $o
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x > $o/t_all.log 2>&1; echo "all rc=$?" | tee -a $o/summary.txt
tail -5 $o/t_all.log
python bench.py > $o/bench_l2hmc.json 2> $o/bench.err; echo "bench rc=$?" | tee -a $o/summary.txt
python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe > $o/bench_train.json 2> $o/bench_train.err; echo "train rc=$?" | tee -a $o/summary.txt
python - $o <<'PY'
import json, sys
o = sys.argv[1]
d = json.loads(open(f'{o}/bench_l2hmc.json').readline())
print('l2hmc', d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'])
print('cpu', d.get('cpu_baseline'))
for k, v in d.get('secondary', {}).items():
    print(' sec', k, v if not isinstance(v, dict) else {a: v[a] for a in ('value', 'ms_per_step', 'accept_prob_mean') if a in v})
for k, v in d.get('secondary_u1', {}).items():
    if isinstance(v, dict):
        print(' u1', k, {a: v[a] for a in ('ms_per_trajectory', 'value', 'default_path', 'eager_instrumented_ms_per_trajectory', 'train_step_s', 'eval_step_s', 'hmc_step_s') if a in v})
    else:
        print(' u1', k, str(v)[:300])
t = json.loads(open(f'{o}/bench_train.json').readline())
print('train', t['value'], t['ms_per_step'], t.get('train'))
PY
tail -3 $o/bench.err
