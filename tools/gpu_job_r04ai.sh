#!/usr/bin/env bash
# round-4 record after the third session: all GPU tests, bench lines (timed region without events), kernel
# tables of the l2hmc and train benches (rocprofv3), cfg-5 shard
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04ai; mkdir -p $o
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > $o/t_all.log 2>&1; echo "all rc=$?" | tee -a $o/summary.txt
tail -4 $o/t_all.log
python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "default bench rc=$?" | tee -a $o/summary.txt
python bench.py --no-cpu-baseline --no-u1 --no-comm-probe > $o/bench_l2hmc.json 2> $o/bench.err; echo "bench rc=$?" | tee -a $o/summary.txt
python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe > $o/bench_train.json 2> $o/bench_train.err; echo "train rc=$?" | tee -a $o/summary.txt
python bench.py --mode hmc --no-u1 --no-cpu-baseline --no-comm-probe > $o/bench_hmc.json 2> $o/bench_hmc.err
python bench.py --lattice 16 16 16 16 --beta 6.2 --warmup 2 --steps 3 --no-cpu-baseline --no-u1 --no-comm-probe --no-spot-check > $o/bench_cfg5_shard.json 2> $o/bench_cfg5.err; echo "cfg5 rc=$?" | tee -a $o/summary.txt
python - $o <<'PY'
import json, sys
o = sys.argv[1]
for f in ('bench_default', 'bench_l2hmc', 'bench_train', 'bench_hmc', 'bench_cfg5_shard'):
    try:
        d = json.loads(open(f'{o}/{f}.json').readline())
    except Exception as e:
        print(f, 'ERR', e); continue
    print(f, d['value'], d['ms_per_step'], d.get('instrumented_ms_per_step'), d.get('kernel_time_fraction_of_wall'))
    for r in d.get('rooflines', []):
        print('     roofline', r.get('kernel')[:40], r.get('frac'), r.get('avg_ms'))
    for k, v in list(d['kernels'].items())[:6]:
        print('   ', k, v)
PY
KSTATS_MARKER=su3_assemble_tah_kernel KSTATS_LAST=5 bash tools/kstats.sh $o/bench_l2hmc_kernel_stats.txt python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe
KSTATS_MARKER=su3_assemble_tah_kernel KSTATS_LAST=5 bash tools/kstats.sh $o/bench_train_kernel_stats.txt python bench.py --mode train --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe
head -14 $o/bench_train_kernel_stats.txt | cut -c1-150
