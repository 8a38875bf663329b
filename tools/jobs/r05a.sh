#!/usr/bin/env bash
# round 5, first pass: the autograd bridge on the GPU, then the whole GPU tier and the bench line
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r05a; mkdir -p $o
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "autograd or bridge or reference_trainer or grad_scaler" > $o/t_bridge.log 2>&1; echo "bridge rc=$?" | tee -a $o/summary.txt
tail -15 $o/t_bridge.log
timeout 2400 python -m pytest tests -q -m gpu > $o/t_all.log 2>&1; echo "all rc=$?" | tee -a $o/summary.txt
tail -15 $o/t_all.log
python bench.py > $o/bench_l2hmc.json 2> $o/bench.err; echo "bench rc=$?" | tee -a $o/summary.txt
python - $o <<'PY'
import json, sys
o = sys.argv[1]
d = json.loads(open(f'{o}/bench_l2hmc.json').readline())
r = d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'][:36], r['frac'], r['traffic'])
for k, v in d.get('rooflines', {}).items():
    print(k, {a: v[a] for a in ('avg_ms', 'frac') if a in v})
PY
