#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
o=gpurun_out/r04aa; mkdir -p $o
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_trainer_gpu.py -q -m gpu -x > $o/t.log 2>&1; echo "train tests rc=$?" | tee -a $o/summary.txt
tail -3 $o/t.log
for ft in 5 2; do
L2Q_TUNE_FORCE_TILE=$ft python - $ft <<'PY' 2>&1 | grep -v amdgpu
import sys, os
sys.path.insert(0, 'l2hmc-qcd_amd'); sys.path.insert(0, 'tools')
import torch
from l2hmc import _ops as ops, native
from kbench import timeit
native.set_tuning('force_tile', int(sys.argv[1]))
nb, L = 256, (8, 8, 8, 8); V = 4096
torch.manual_seed(0)
xn = ops.su3_project_su_n(torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda'))
gf = torch.randn(nb, 4, 9, V, dtype=torch.complex128, device='cuda')
gx = torch.zeros_like(xn)
t = timeit(lambda: native.call('l2q_su3_force_bwd', xn, gf, 6.0, gx, nb, *L), iters=10, warm=3)
print(f'force_tile={sys.argv[1]}: l2q_su3_force_bwd 8^4 x 256: {t*1e3:.4f} ms')
PY
done
for i in 1 2; do
python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe > $o/bench_train_$i.json 2> $o/bench_train.err; echo "train rc=$?" | tee -a $o/summary.txt
python - $o/bench_train_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('instrumented_ms_per_step'))
for k, v in list(d['kernels'].items())[:9]:
    print('   ', k, v)
PY
done
