#!/usr/bin/env bash
# round 4, first pass: the new parity tests (from-seed trajectories, cfg-4 at 256 chains), then the
# whole GPU tier and the bench line at HEAD
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04a; mkdir -p $o
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dynamics_gpu.py -q -m gpu -k "from_seed" > $o/t_seed.log 2>&1; echo "from_seed rc=$?" | tee -a $o/summary.txt
timeout 900 python -m pytest tests/test_sizes_gpu.py -q -m gpu -k "cfg4" > $o/t_cfg4.log 2>&1; echo "cfg4 rc=$?" | tee -a $o/summary.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $o/t_all.log 2>&1; echo "all rc=$?" | tee -a $o/summary.txt
python bench.py > $o/bench_l2hmc.json 2> $o/bench.err; echo "bench rc=$?" | tee -a $o/summary.txt
tail -5 $o/t_seed.log; tail -5 $o/t_cfg4.log; tail -3 $o/t_all.log
python - $o <<'PY'
import json, sys
o = sys.argv[1]
d = json.loads(open(f'{o}/bench_l2hmc.json').readline())
r = d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'][:36], r['frac'], r['traffic'])
for k, v in d.get('rooflines', {}).items():
    print(k, v)
PY
