#!/usr/bin/env bash
# A/B of this session's kernel changes on one box: LDS chunk swizzle of the fp64 MFMA layers,
# x-update register budget, split-K reduce; then the parity tests that cover them.
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r03d; mkdir -p $o
for lib in libl2q.so libl2q_oldswz.so; do
  echo "## $lib"; L2Q_LIB_NAME=$lib python tools/bench_heads_f64.py 2>&1 | grep -v amdgpu.ids
  L2Q_LIB_NAME=$lib python tools/kbench.py --gemm 2>&1 | grep -iE "gemm|heads|expm|projsu"
done > $o/heads_swizzle_ab.txt 2>&1
for lib in libl2q.so libl2q_xu4.so libl2q_xu2.so; do
  echo "## $lib"; L2Q_LIB_NAME=$lib python tools/kbench.py 2>&1 | grep -iE "expm"
done > $o/expm_occ_ab.txt 2>&1
for lib in libl2q.so libl2q_oldswz.so libl2q.so; do
  echo "## $lib"; L2Q_LIB_NAME=$lib python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step']); [print('  ',k,v['avg_ms'],v['launches']) for k,v in list(d['kernels'].items())[:7]]"
done > $o/bench_ab.txt 2>&1
python -m pytest tests -x -q -m gpu -k "gemm or heads or expm or su3 or cfg4 or cfg5_kernels" > $o/tests.log 2>&1; tail -3 $o/tests.log
mkdir -p tools/bin; hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_b64_conflict.hip -o tools/bin/lds_b64_conflict 2>/dev/null
d=$o/lds_micro; mkdir -p $d; R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$d -o p --output-format csv -- $R/tools/bin/lds_b64_conflict 4000 > $R/$d/stdout.log 2>&1)
python - $d > $o/lds_b64_conflict.txt <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print('\n'.join(l for l in open(sys.argv[1] + '/stdout.log').read().splitlines() if l.startswith('mode')))
for k, v in sorted(agg.items()):
    print(k, {c: sum(x) / len(x) for c, x in v.items()})
PY
cat $o/heads_swizzle_ab.txt $o/expm_occ_ab.txt $o/bench_ab.txt $o/lds_b64_conflict.txt
