#!/usr/bin/env bash
# round-2 measurement job: force A/B incl. 16^4, bench (eval/train/hmc), rocprof kernel stats, PMC
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/f
o=gpurun_out/f
export TMPDIR=/tmp
timeout 600 python tools/force_bench.py --big > $o/force_bench.log 2>&1
timeout 600 python bench.py > $o/bench.json 2> $o/bench.err; echo "rc=$?" >> $o/bench.err
timeout 600 python bench.py --mode hmc --no-cpu-baseline > $o/bench_hmc.json 2>> $o/bench.err
timeout 900 python bench.py --mode train --steps 3 --warmup 1 > $o/bench_train.json 2>> $o/bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$o/prof -o r --output-format csv -- python $OLDPWD/bench.py --no-cpu-baseline --no-spot-check > $OLDPWD/$o/bench_under_rocprof.json 2> $OLDPWD/$o/rocprof.err)
python - <<'PY' > gpurun_out/f/kernel_stats.txt
import csv, glob, re
f = glob.glob('gpurun_out/f/prof/**/r_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
print('# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-spot-check  (MI355X)')
print(f'{"kernel":100s} {"calls":>6s} {"total_ms":>10s} {"avg_us":>10s} {"%":>6s}')
for r in rows[:45]:
    n = re.sub(r'\(.*', '', r['Name']).replace('void ', '')[:100]
    print(f'{n:100s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e6:10.3f} {float(r["AverageNs"])/1e3:10.2f} {float(r["Percentage"]):6.2f}')
PY
bash tools/pmc_collect.sh r02d > $o/pmc.log 2>&1
cp profiles/r02d_pmc_counters.txt profiles/pmc_traffic.json $o/ 2>/dev/null
cat $o/force_bench.log | grep "force_tile=[24]"; head -c 200 $o/bench.json; echo; head -12 $o/kernel_stats.txt
