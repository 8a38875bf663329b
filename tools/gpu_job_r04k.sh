#!/usr/bin/env bash
# bench line with the int8-sliced input layer (+ spot check), the fp64 A/B, kernel table
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04k; mkdir -p $o
export TMPDIR=/tmp
python bench.py > $o/bench_l2hmc.json 2> $o/bench.err; echo "bench rc=$?" | tee -a $o/summary.txt
python bench.py --fp64-input-layer --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > $o/bench_l2hmc_fp64_input.json 2> $o/bench2.err; echo "bench (fp64 input) rc=$?" | tee -a $o/summary.txt
python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > $o/bench_l2hmc_b.json 2> $o/bench3.err; echo "bench (again) rc=$?" | tee -a $o/summary.txt
tail -3 $o/bench.err
python - $o <<'PY'
import json, sys
o = sys.argv[1]
for f in ('bench_l2hmc.json', 'bench_l2hmc_fp64_input.json', 'bench_l2hmc_b.json'):
    try:
        d = json.loads(open(f'{o}/{f}').readline())
    except Exception as e:
        print(f, 'unreadable', e); continue
    r = d['roofline']
    print(f, d['value'], d['ms_per_step'], r['kernel'][:36], r['frac'], r['traffic'])
    for r in d.get('rooflines', []):
        print('   ', r['kernel'][:50], r['avg_ms'], r['frac'], r.get('int8', {}).get('frac'))
    if 'spot_check' in d:
        print('   ', d['spot_check'].get('input_layer'))
    print('   ', d.get('secondary'))
PY
