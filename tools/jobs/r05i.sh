#!/usr/bin/env bash
# round 5: SU(3) auto-graph -- dynamics / trainer / sizes tests, then the bench line (default flags) and hmc
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r05i; mkdir -p $o
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x > $o/t_all.log 2>&1; echo "all rc=$?" | tee -a $o/summary.txt
tail -8 $o/t_all.log
python bench.py --no-u1 > $o/bench_l2hmc.json 2> $o/bench.err; echo "bench rc=$?" | tee -a $o/summary.txt
python bench.py --mode hmc --no-u1 --no-cpu-baseline > $o/bench_hmc.json 2>> $o/bench.err; echo "hmc rc=$?" | tee -a $o/summary.txt
python bench.py --no-u1 --no-cpu-baseline --settle 0 > $o/bench_l2hmc_cold.json 2>> $o/bench.err
python - $o <<'PY'
import json, sys
o = sys.argv[1]
for f in ('bench_l2hmc', 'bench_hmc', 'bench_l2hmc_cold'):
    d = json.loads(open(f'{o}/{f}.json').readline())
    print(f, d['value'], d['ms_per_step'], d['instrumented_ms_per_step'], d['setup_steps'], d['launch_path'][:30], d['roofline']['avg_ms'], d['roofline']['frac'])
    for k, v in d.get('secondary', {}).items():
        print('   sec', k, v if not isinstance(v, dict) else v.get('value'))
PY
tail -3 $o/bench.err
