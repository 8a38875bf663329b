#!/usr/bin/env bash
# mid-round record: full GPU tier, bench line (new secondary lines), traffic counters incl. the pair force kernel
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04g; mkdir -p $o
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > $o/t_all.log 2>&1; echo "all rc=$?" | tee -a $o/summary.txt
python bench.py > $o/bench_l2hmc.json 2> $o/bench.err; echo "bench rc=$?" | tee -a $o/summary.txt
bash tools/pmc_collect.sh r04g > $o/pmc.log 2>&1
cp profiles/r04g_pmc_counters.txt profiles/pmc_traffic.json $o/ 2>/dev/null
tail -6 $o/t_all.log
python - $o <<'PY'
import json, sys
o = sys.argv[1]
d = json.loads(open(f'{o}/bench_l2hmc.json').readline())
r = d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'][:36], r['frac'], r['traffic'])
for r in d.get('rooflines', []):
    print(r['kernel'][:50], r['avg_ms'], r['frac'])
print(d['secondary'])
PY
grep -A 26 "su3_force_pair_kernel<0" profiles/r04g_pmc_counters.txt | tail -4; grep -A 26 "su3_force_link_kernel<0" profiles/r04g_pmc_counters.txt | tail -4
