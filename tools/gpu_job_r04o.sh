#!/usr/bin/env bash
# after the LDS swizzle of gemm_sliced: bench line, kernel table of the timed region, PMC passes
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04o; mkdir -p $o
export TMPDIR=/tmp
python bench.py > $o/bench_l2hmc.json 2> $o/bench.err; echo "bench rc=$?" | tee -a $o/summary.txt
KSTATS_MARKER=su3_assemble_tah_kernel KSTATS_LAST=5 bash tools/kstats.sh "$o/bench_l2hmc_kernel_stats.txt" python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/kstats_l2hmc.log" 2>&1
bash tools/pmc_collect.sh r04o > $o/pmc.log 2>&1
cp profiles/r04o_pmc_counters.txt profiles/pmc_traffic.json $o/ 2>/dev/null
python bench.py --lattice 16 16 16 16 --beta 6.2 --steps 3 --warmup 2 --no-cpu-baseline --no-spot-check --no-u1 > "$o/bench_cfg5_shard.json" 2> "$o/bench_cfg5.err"
head -8 $o/bench_l2hmc_kernel_stats.txt
grep -A26 "gemm_sliced_kernel" $o/r04o_pmc_counters.txt | grep "LDS\|traffic\|hit"
python - $o <<'PY'
import json, sys
o = sys.argv[1]
for f in ('bench_l2hmc', 'bench_cfg5_shard'):
    d = json.loads(open(f'{o}/{f}.json').readline())
    print(f, d['value'], d['ms_per_step'])
    for r in d.get('rooflines', []):
        print('   ', r['kernel'][:50], r['avg_ms'], r['frac'], r.get('int8', {}).get('frac'), r.get('traffic'))
    print('   ', d.get('secondary'))
    if 'secondary_u1' in d:
        for k, v in d['secondary_u1'].items():
            print('   ', k, v if isinstance(v, str) else (v['ms_per_trajectory'], v['value'], v.get('hip_graph')))
PY
