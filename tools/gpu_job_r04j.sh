#!/usr/bin/env bash
# int8-sliced input layer in the trajectory: full GPU tier, bench line with the new roofline entry and the
# input-layer spot check, and the same bench with the fp64 input layer for comparison
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04j; mkdir -p $o
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > $o/t_all.log 2>&1; echo "all rc=$?" | tee -a $o/summary.txt
python bench.py > $o/bench_l2hmc.json 2> $o/bench.err; echo "bench rc=$?" | tee -a $o/summary.txt
python bench.py --fp64-input-layer --no-cpu-baseline --no-spot-check > $o/bench_l2hmc_fp64_input.json 2> $o/bench2.err; echo "bench (fp64 input) rc=$?" | tee -a $o/summary.txt
tail -6 $o/t_all.log
python - $o <<'PY'
import json, sys
o = sys.argv[1]
for f in ('bench_l2hmc.json', 'bench_l2hmc_fp64_input.json'):
    d = json.loads(open(f'{o}/{f}').readline())
    r = d['roofline']
    print(f, d['value'], d['ms_per_step'], r['kernel'][:36], r['frac'], r['traffic'])
    for r in d.get('rooflines', []):
        print('   ', r['kernel'][:50], r['avg_ms'], r['frac'], r.get('int8', {}).get('frac'))
    if 'spot_check' in d:
        print('   ', d['spot_check'].get('input_layer'))
    print('   ', d.get('secondary'))
PY
