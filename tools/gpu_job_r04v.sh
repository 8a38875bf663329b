#!/usr/bin/env bash
# round-4 session 3 record: all GPU tests, bench lines (two-pass timing), kernel table of the l2hmc bench
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04v; mkdir -p $o
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x > $o/t_all.log 2>&1; echo "all rc=$?" | tee -a $o/summary.txt
tail -4 $o/t_all.log
python bench.py --no-cpu-baseline --no-u1 --no-comm-probe > $o/bench_l2hmc.json 2> $o/bench.err; echo "bench rc=$?" | tee -a $o/summary.txt
python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe > $o/bench_train.json 2> $o/bench_train.err; echo "train rc=$?" | tee -a $o/summary.txt
python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe --fp64-train-heads > $o/bench_train_fp64heads.json 2>> $o/bench_train.err
python bench.py --mode hmc --no-u1 --no-cpu-baseline --no-comm-probe > $o/bench_hmc.json 2> $o/bench_hmc.err
python - $o <<'PY'
import json, sys
o = sys.argv[1]
for f in ('bench_l2hmc', 'bench_train', 'bench_train_fp64heads', 'bench_hmc'):
    d = json.loads(open(f'{o}/{f}.json').readline())
    print(f, d['value'], d['ms_per_step'], d.get('instrumented_ms_per_step'), d.get('kernel_time_fraction_of_wall'))
    for k, v in list(d['kernels'].items())[:8]:
        print('   ', k, v)
PY
KSTATS_MARKER=su3_assemble_tah_kernel KSTATS_LAST=5 bash tools/kstats.sh $o/bench_l2hmc_kernel_stats.txt python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe
head -12 $o/bench_l2hmc_kernel_stats.txt | cut -c1-150
