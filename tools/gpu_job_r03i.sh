#!/usr/bin/env bash
# cfg-5 PMC pass (16^4 x 256 chains) incl. the sliced heads kernel, then the bench lines at HEAD
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r03i; mkdir -p $o
export TMPDIR=/tmp
L2Q_KPROF_LATTICE="16 16 16 16" L2Q_KPROF_NB=256 bash tools/pmc_collect.sh r03h_16x4 > $o/pmc16.log 2>&1
cp profiles/r03h_16x4_pmc_counters.txt profiles/pmc_traffic.json $o/
python bench.py > $o/bench_l2hmc.json 2> $o/bench.err
python bench.py --lattice 16 16 16 16 --beta 6.2 --steps 3 --warmup 2 --no-cpu-baseline --no-spot-check --no-u1 > $o/bench_cfg5_shard.json 2>> $o/bench.err
python bench.py --mode train --no-u1 > $o/bench_train.json 2>> $o/bench.err
tail -3 $o/pmc16.log
grep -A 24 "heads_sliced_kernel<true, true, true, false>" profiles/r03h_16x4_pmc_counters.txt | tail -6
python - $o <<'PY'
import json, sys
o = sys.argv[1]
for f in ('bench_l2hmc', 'bench_cfg5_shard', 'bench_train'):
    try:
        d = json.loads(open(f'{o}/{f}.json').readline())
        r = d['roofline']
        print(f, d['value'], d['ms_per_step'], r['kernel'][:36], r['frac'], r['traffic'], r.get('int8'))
    except Exception as e:
        print(f, 'failed', e)
PY
