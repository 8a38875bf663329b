#!/usr/bin/env bash
# Round-3 measurement record at HEAD (run ON the GPU box through gpurun):
#   kernel-trace --stats summaries of bench.py (cfg-4 l2hmc / hmc) and of the cfg-5 shard,
#   PMC passes at 8^4 and at 16^4, force A/B incl. 16^4, LDS b64 microbenchmark under PMC.
set -u
cd "$(dirname "$0")/.."
tag="${1:-r03b}"
o="gpurun_out/$tag"; mkdir -p "$o"
export TMPDIR=/tmp
python bench.py > "$o/bench_l2hmc.json" 2> "$o/bench_l2hmc.err"
python bench.py --mode hmc --no-u1 > "$o/bench_hmc.json" 2>> "$o/bench_l2hmc.err"
bash tools/kstats.sh "$o/bench_l2hmc_kernel_stats.txt" python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/kstats_l2hmc.log" 2>&1
bash tools/kstats.sh "$o/bench_hmc_kernel_stats.txt" python bench.py --mode hmc --no-cpu-baseline --no-spot-check --no-u1 > "$o/kstats_hmc.log" 2>&1
# cfg-5 per-GPU shard: 16^4, 256 chains, beta 6.2
python bench.py --lattice 16 16 16 16 --beta 6.2 --steps 2 --warmup 1 --no-cpu-baseline --no-spot-check --no-u1 > "$o/bench_cfg5_shard.json" 2> "$o/bench_cfg5.err"
bash tools/kstats.sh "$o/bench_cfg5_shard_kernel_stats.txt" python bench.py --lattice 16 16 16 16 --beta 6.2 --steps 2 --warmup 1 --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/kstats_cfg5.log" 2>&1
python tools/force_bench.py --big > "$o/force_variants_ab.txt" 2>&1
bash tools/pmc_collect.sh "${tag}" > "$o/pmc_8x4.log" 2>&1
L2Q_KPROF_LATTICE="16 16 16 16" L2Q_KPROF_NB=256 bash tools/pmc_collect.sh "${tag}_16x4" > "$o/pmc_16x4.log" 2>&1
cp profiles/${tag}_pmc_counters.txt profiles/${tag}_16x4_pmc_counters.txt profiles/pmc_traffic.json "$o/" 2>/dev/null
mkdir -p tools/bin; hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_b64_conflict.hip -o tools/bin/lds_b64_conflict 2>/dev/null
d="$o/lds_micro"; mkdir -p "$d"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$OLDPWD/$d" -o p --output-format csv -- "$OLDPWD/tools/bin/lds_b64_conflict" 4000 > "$OLDPWD/$d/stdout.log" 2>&1)
python - "$d" > "$o/lds_b64_conflict.txt" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + '/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:40]].setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
print(open(sys.argv[1] + '/stdout.log').read())
for k, v in sorted(agg.items()):
    print(k, {c: sum(x) / len(x) for c, x in v.items()})
PY
ls "$o"
