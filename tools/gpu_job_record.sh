#!/usr/bin/env bash
# Round record (tag = first argument): full GPU suite, bench lines, kernel tables of the timed region, PMC passes.
set -u
cd "$(dirname "$0")/.."
tag="${1:-r04m}"
o="gpurun_out/$tag"; mkdir -p "$o"
export TMPDIR=/tmp
M=su3_assemble_tah_kernel
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee "$o/pytest_gpu.txt"
python bench.py > "$o/bench_l2hmc.json" 2> "$o/bench_l2hmc.err"
python bench.py --mode hmc --no-u1 > "$o/bench_hmc.json" 2>> "$o/bench_l2hmc.err"
python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe > "$o/bench_train.json" 2> "$o/bench_train.err"
python bench.py --lattice 16 16 16 16 --beta 6.2 --steps 3 --warmup 2 --no-cpu-baseline --no-spot-check --no-u1 > "$o/bench_cfg5_shard.json" 2> "$o/bench_cfg5.err"
KSTATS_MARKER=$M KSTATS_LAST=5 bash tools/kstats.sh "$o/bench_l2hmc_kernel_stats.txt" python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/kstats_l2hmc.log" 2>&1
KSTATS_MARKER=$M KSTATS_LAST=3 bash tools/kstats.sh "$o/bench_cfg5_shard_kernel_stats.txt" python bench.py --lattice 16 16 16 16 --beta 6.2 --steps 3 --warmup 2 --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/kstats_cfg5.log" 2>&1
bash tools/pmc_collect.sh "$tag" > "$o/pmc.log" 2>&1
cp profiles/${tag}_pmc_counters.txt profiles/pmc_traffic.json "$o/" 2>/dev/null
python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/bench_l2hmc_after_pmc.json" 2>/dev/null
head -16 "$o/bench_l2hmc_kernel_stats.txt"; head -10 "$o/bench_cfg5_shard_kernel_stats.txt"
python - "$o" <<'PY'
import json, sys
o = sys.argv[1]
for f in ('bench_l2hmc', 'bench_hmc', 'bench_cfg5_shard', 'bench_l2hmc_after_pmc'):
    try:
        d = json.loads(open(f'{o}/{f}.json').readline())
        print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'][:40], d['roofline']['frac'], d['roofline']['traffic'])
    except Exception as e:
        print(f, 'failed', e)
PY
